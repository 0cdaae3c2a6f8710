// gmm_kernels.hip -- GMM scoring and the HMM recursion on device-resident MFCC vectors (gfx950).
//
//   gmm_score_kernel     Recognition() + the class arg-max   GMMAlgorithm_Test_Auto_ver2.cpp:113-127,:151-162
//   gmm_emission_kernel  log sum_k alpa[k] probability_k(x)   Viterbi_version1.cpp:183-186,:193-196
//   hmm_trellis_kernel   HMMRecognition()                      Viterbi_version1.cpp:157-246
//
// Everything is FP64 like the reference (these are a few hundred flops per 96-byte vector: neither HBM nor
// the FP64 pipe is stressed; the point of the device path is that the MFCCs never leave HBM).  A GMM's
// parameters are wave-uniform (one class per wave), so they come in through scalar loads.
#include "jdsp_internal.h"

namespace jdsp {

// probability() (GMMTest:216-235 = Viterbi:248-267) for mixture k of the packed record `g`, and the
// alpa-weighted sum over the four mixtures.  No FMA contraction: the products and sums round one by one as
// in the reference's x87-free double arithmetic.
__device__ __forceinline__ double gmm_mixture(const double (&x)[12], const double *__restrict__ g)
{
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double *E = g + kGmmEig + 48 * k;
        double y[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 12; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) y[j] = __dadd_rn(y[j], __dmul_rn(x[i], E[4 * i + j]));      // :228
        }
        double p = 1.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {                                                               // :230-233
            const double c = g[kGmmVar + 4 * k + i];
            const double d = __dsub_rn(y[i], g[kGmmMean + 4 * k + i]);
            const double e = exp(__ddiv_rn(__dmul_rn(-0.5, __dmul_rn(d, d)), c));
            p = __dmul_rn(p, __dmul_rn(g[kGmmCoef + 4 * k + i], e));
        }
        t = __dadd_rn(t, __dmul_rn(g[kGmmAlpa + k], p));                                            // :156
    }
    return t;
}

// The same density with the arithmetic fused: FMA projections, -0.5/var precomputed, ONE exp per mixture
// (the product of four exponentials is the exponential of the sum) -- about a third of the instructions.
// Equal to the reference's order to a few 1e-16 relative, except where a mixture density is itself denormal
// (< 2.2e-308), where both forms have already lost their digits.  Opt-in: jdsp_gmm_set_option "evaluation".
__device__ __forceinline__ double gmm_mixture_fused(const double (&x)[12], const double *__restrict__ g)
{
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const double *E = g + kGmmEig + 48 * k;
        double y[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 12; i++) {
#pragma unroll
            for (int j = 0; j < 4; j++) y[j] = fma(x[i], E[4 * i + j], y[j]);
        }
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const double d = y[i] - g[kGmmMean + 4 * k + i];
            s = fma(d * d, g[kGmmNhiv + 4 * k + i], s);
        }
        t = fma(g[kGmmAlpa + k] * g[kGmmCprod + k], exp(s), t);
    }
    return t;
}

__device__ __forceinline__ void load_vector(const double *__restrict__ feats, long f, double (&x)[12])
{
    const double2 *p = reinterpret_cast<const double2 *>(feats + 12 * f);                          // 96 B, 16-aligned
#pragma unroll
    for (int i = 0; i < 6; i++) {
        const double2 v = p[i];
        x[2 * i] = v.x;
        x[2 * i + 1] = v.y;
    }
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// One workgroup (4 waves) per utterance; wave w scores the classes w, w+4, ...: lanes stride over the
// utterance's frames, a fixed butterfly adds the 64 partial sums.  Thread 0 then runs the reference's
// `dMax < score` scan.  The sum over frames is therefore associated differently from the reference's
// running sum (O(1e-16) relative per term).
__device__ __forceinline__ long clamp_offset(long long v, long n) { return v < 0 ? 0 : (v > n ? n : (long)v); }

template <bool FUSED>
__global__ __launch_bounds__(256) void gmm_score_kernel(const double *__restrict__ feats, long n_frames,
                                                        const long long *__restrict__ utt_first, long n_utts,
                                                        const double *__restrict__ gmm, int n_classes,
                                                        double *__restrict__ scores, int *__restrict__ best)
{
    __shared__ double sc[kGmmMaxClasses];
    const long u = blockIdx.x;
    if (u >= n_utts) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // offsets outside [0, n_frames] are the caller's error; clamped so that no vector outside feats is read
    const long first = clamp_offset(utt_first[u], n_frames), last = clamp_offset(utt_first[u + 1], n_frames);
    for (int c = wave; c < n_classes; c += 4) {
        const double *g = gmm + (size_t)c * kGmmRecord;
        double acc = 0.0;
        for (long f = first + lane; f < last; f += 64) {
            double x[12];
            load_vector(feats, f, x);
            acc += log(FUSED ? gmm_mixture_fused(x, g) : gmm_mixture(x, g));                        // :158
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            const double s = acc / (double)(last - first);                                          // :161
            sc[c] = s;
            scores[u * n_classes + c] = s;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0 && best) {
        double mx = sc[0];
        int arg = 0;
        for (int c = 1; c < n_classes; c++)
            if (mx < sc[c]) { mx = sc[c]; arg = c; }                                                // :117-124
        best[u] = arg;
    }
}

// log b[f][g] for every frame f and every state GMM g (n_g = models * 6); 64 frames x one g per wave.
template <bool FUSED>
__global__ __launch_bounds__(64) void gmm_emission_kernel(const double *__restrict__ feats, long n_frames,
                                                          const double *__restrict__ gmm, int n_g,
                                                          double *__restrict__ b)
{
    const long f = (long)blockIdx.x * 64 + threadIdx.x;
    const int gi = blockIdx.y;
    if (f >= n_frames) return;
    double x[12];
    load_vector(feats, f, x);
    const double *g = gmm + (size_t)gi * kGmmRecord;
    b[f * n_g + gi] = log(FUSED ? gmm_mixture_fused(x, g) : gmm_mixture(x, g));                     // Viterbi:186,:196
}

// Eight lanes per utterance (six used, one per state m), models in turn: the six-state recursion as the
// reference writes it (log of the previous accumulated log probability, :196; first-u assignment then `<`,
// :198-204), the per-frame arg-max state (:212-221) as the "decoding result", the frame-1 maximum as the
// score, and the model arg-max of :119-126.  `lb` holds log(b) from gmm_emission_kernel, so the serial chain
// per vector is one log(), six exchanges within the group and the compare ladder.
__global__ __launch_bounds__(64) void hmm_trellis_kernel(const double *__restrict__ lb,
                                                         const long long *__restrict__ utt_first, long n_utts,
                                                         long n_frames_total, const double *__restrict__ log_trans,
                                                         int n_models, double log_init, double *__restrict__ scores,
                                                         int *__restrict__ best, int *__restrict__ path,
                                                         double *__restrict__ trellis)
{
    const int lane = threadIdx.x;
    const int base = lane & ~7, m = (lane & 7) < 6 ? (lane & 7) : 5;   // lanes 6, 7 of a group shadow state 5
    const bool writer = (lane & 7) < 6, leader = (lane & 7) == 0;
    long u = (long)blockIdx.x * 8 + (lane >> 3);
    const bool live = u < n_utts;
    if (!live) u = n_utts - 1;                                          // keep the whole wave in the exchanges
    const long first = clamp_offset(utt_first[u], n_frames_total), last = clamp_offset(utt_first[u + 1], n_frames_total);
    // every group of the wave walks as many vectors as the longest one (exchanges are wave-wide)
    long len = last - first;
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        const long other = __shfl_xor(len, o, 64);
        len = other > len ? other : len;
    }
    const int n_g = n_models * 6;
    double mx_model = 0.0;
    int arg_model = 0;
    for (int mdl = 0; mdl < n_models; mdl++) {
        double lt[6];
#pragma unroll
        for (int q = 0; q < 6; q++) lt[q] = log_trans[36 * mdl + 6 * q + m];                        // into state m
        double cur = 0.0, ret = 0.0;
        for (long i = 0; i < len; i++) {
            const long f = first + i;
            const bool on = f < last;
            const double lbm = on ? lb[f * n_g + 6 * mdl + m] : 0.0;
            const double lp = log(cur);                                                             // :196
            double nxt;
            if (i == 0) {
                nxt = lbm + log_init;                                                               // :186
            } else {
                nxt = (__shfl(lp, base, 64) + lt[0]) + lbm;                                         // u = 0
#pragma unroll
                for (int q = 1; q < 6; q++) {
                    const double t = (__shfl(lp, base + q, 64) + lt[q]) + lbm;
                    if (nxt < t) nxt = t;                                                           // :201
                }
            }
            double mx = __shfl(nxt, base, 64);
            int arg = 0;
#pragma unroll
            for (int q = 1; q < 6; q++) {
                const double c = __shfl(nxt, base + q, 64);
                if (c > mx) { mx = c; arg = q; }                                                    // :217
            }
            if (on) {
                cur = nxt;
                if (i == 1) ret = mx;
                if (live && leader && path) path[(size_t)mdl * n_frames_total + f] = i == 0 ? 0 : arg;
                if (live && writer && trellis) trellis[((size_t)mdl * 6 + m) * n_frames_total + f] = nxt;
            }
        }
        if (live && leader && scores) scores[u * n_models + mdl] = ret;
        if (mdl == 0) { mx_model = ret; arg_model = 0; }
        else if (mx_model < ret) { mx_model = ret; arg_model = mdl; }                               // Viterbi:119-126
    }
    if (live && leader && best) best[u] = arg_model;
}

int launch_gmm_score(hipStream_t stream, const double *feats, long n_frames, const long long *utt_first, long n_utts,
                     const double *gmm, int n_classes, int fused, double *scores, int *best)
{
    if (n_utts <= 0) return 0;
    if (fused)
        hipLaunchKernelGGL(gmm_score_kernel<true>, dim3((unsigned)n_utts), dim3(256), 0, stream, feats, n_frames, utt_first,
                           n_utts, gmm, n_classes, scores, best);
    else
        hipLaunchKernelGGL(gmm_score_kernel<false>, dim3((unsigned)n_utts), dim3(256), 0, stream, feats, n_frames, utt_first,
                           n_utts, gmm, n_classes, scores, best);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

int launch_hmm_viterbi(hipStream_t stream, const double *feats, long n_frames, const long long *utt_first, long n_utts,
                       const double *gmm, const double *log_trans, int n_models, int fused, double log_init, double *b,
                       double *scores, int *best, int *path, double *trellis)
{
    if (n_utts <= 0) return 0;
    if (n_frames > 0) {
        const dim3 grid((unsigned)((n_frames + 63) / 64), (unsigned)(n_models * 6));
        if (fused) hipLaunchKernelGGL(gmm_emission_kernel<true>, grid, dim3(64), 0, stream, feats, n_frames, gmm, n_models * 6, b);
        else hipLaunchKernelGGL(gmm_emission_kernel<false>, grid, dim3(64), 0, stream, feats, n_frames, gmm, n_models * 6, b);
        if (hipGetLastError() != hipSuccess) return -1;
    }
    hipLaunchKernelGGL(hmm_trellis_kernel, dim3((unsigned)((n_utts + 7) / 8)), dim3(64), 0, stream, b, utt_first,
                       n_utts, n_frames, log_trans, n_models, log_init, scores, best, path, trellis);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}

}  // namespace jdsp
