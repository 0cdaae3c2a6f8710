// stage_api.hip -- the reference's sub-step functions as separately callable ABI entries (SURVEY §8b lists them
// among the signatures to keep): MelFilterBank / DCT / Liftering of MFCCFeatureExtraction_auto_version1.cpp:154-192
// and probability() of GMMAlgorithm_Test_Auto_ver2.cpp:164-236 (= Viterbi_version1.cpp:248-267).
//
// The batched chains (mfcc_kernels.hip, gmm_kernels.hip) fuse these steps; a caller that keeps the reference's
// function structure -- its own |X| loop in front of MelFilterBank, its own loop around probability() -- gets them
// here, in FP64 like the reference and evaluated in the reference's operation order (sums in bin / channel order,
// products and sums rounded one by one), for any number of rows at once.  Trigonometric constants come from
// host-built tables (the same libm the reference links), so only log() and exp() can differ from a CPU run, in
// their last place.
#include "jdsp_internal.h"

using jdsp::fail;

namespace jdsp {

// MFCC:154-174.  One thread per (row, channel): walks the bins in order and adds what the reference adds to this
// channel -- (1 - fb[i]) |X[i]| for bins whose rgdFiBins value is the channel (:160,:165), fb[i] |X[i]| for bins
// whose value is the channel + 1 (:163) -- so every channel's sum is formed in the reference's order.  Then ln.
__global__ __launch_bounds__(64) void mel_filterbank_f64_kernel(const double *__restrict__ mag, long n_rows, int n_bins,
                                                               int n_chan, const int *__restrict__ fi,
                                                               const double *__restrict__ fb, double *__restrict__ mel)
{
    const long row = blockIdx.x;
    const int ch = threadIdx.x;
    if (row >= n_rows || ch >= n_chan) return;
    const double *a = mag + row * n_bins;
    double m = 0.0;
    for (int i = 0; i < n_bins; i++) {
        const int k = fi[i];
        if (k == ch) m = __dadd_rn(m, __dmul_rn(__dsub_rn(1.0, fb[i]), a[i]));             // k == 0 (:160) or k != CHANNEL (:165)
        else if (k == ch + 1) m = __dadd_rn(m, __dmul_rn(fb[i], a[i]));                    // :163
    }
    mel[row * n_chan + ch] = log(m);                                                       // :170-172
}

// MFCC:176-183: cep[i-1] += sqrt(2/C) * mel[k-1] * cos(PI i (k-0.5)/C), k ascending, ADDED to what cep holds.
__global__ __launch_bounds__(32) void dct_f64_kernel(const double *__restrict__ mel, long n_rows, int n_chan, int n_cep,
                                                    double scale, const double *__restrict__ cosv, double *__restrict__ cep)
{
    const long row = blockIdx.x;
    const int i = threadIdx.x;
    if (row >= n_rows || i >= n_cep) return;
    const double *m = mel + row * n_chan;
    double acc = cep[row * n_cep + i];
    for (int k = 0; k < n_chan; k++) acc = __dadd_rn(acc, __dmul_rn(__dmul_rn(scale, m[k]), cosv[(size_t)i * n_chan + k]));
    cep[row * n_cep + i] = acc;
}

// MFCC:185-192
__global__ __launch_bounds__(32) void lifter_f64_kernel(double *__restrict__ cep, long n_rows, int n_cep,
                                                       const double *__restrict__ lift)
{
    const long row = blockIdx.x;
    const int i = threadIdx.x;
    if (row >= n_rows || i >= n_cep) return;
    cep[row * n_cep + i] = __dmul_rn(cep[row * n_cep + i], lift[i]);
}

// probability() (GMMTest:216-235): y = x E (in-order sums over the 12 inputs), product over the 4 axes of
// (1/sqrt(2 PI)) (1/sqrt(c_i)) exp((-1/2) (y_i - mean_i)^2 / c_i).  par: mean[4], var[4], coef[4], eig[12][4].
__global__ __launch_bounds__(256) void gmm_probability_kernel(const double *__restrict__ feats, long n,
                                                              const double *__restrict__ par, double *__restrict__ prob)
{
    const long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= n) return;
    const double *x = feats + 12 * v;
    double y[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < 12; i++)
        for (int j = 0; j < 4; j++) y[j] = __dadd_rn(y[j], __dmul_rn(x[i], par[12 + 4 * i + j]));      // :228
    double p = 1.0;
    for (int i = 0; i < 4; i++) {                                                                       // :230-233
        const double c = par[4 + i];
        const double d = __dsub_rn(y[i], par[i]);
        const double e = exp(__ddiv_rn(__dmul_rn(-0.5, __dmul_rn(d, d)), c));
        p = __dmul_rn(p, __dmul_rn(par[8 + i], e));
    }
    prob[v] = p;
}

}  // namespace jdsp

namespace {

// host -> device -> host round trip shared by the four entries
struct Staged {
    jdsp_ctx *ctx;
    void *d[3] = {nullptr, nullptr, nullptr};
    explicit Staged(jdsp_ctx *c) : ctx(c) {}
    ~Staged()
    {
        for (void *p : d)
            if (p) (void)hipFree(p);
    }
    int up(int slot, const void *host, size_t bytes)
    {
        hipError_t e = hipMalloc(&d[slot], bytes ? bytes : 1);
        if (e != hipSuccess) return fail(ctx, JDSP_ENOMEM, "stage: hipMalloc", e);
        if (host && (e = hipMemcpyAsync(d[slot], host, bytes, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
            return fail(ctx, JDSP_EHIP, "stage: H2D", e);
        return JDSP_OK;
    }
    int down(int slot, void *host, size_t bytes)
    {
        hipError_t e = hipMemcpyAsync(host, d[slot], bytes, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) return fail(ctx, JDSP_EHIP, "stage: D2H", e);
        return JDSP_OK;
    }
};

// FP64 tables of the sub-step entries, built on first use: rgdFiBins / rgdFilterBank as MelFilterBankInit left
// them (:131-150), cos(PI i (k-0.5)/C) of :180 and the lifter weights of :189 from the host's libm.
int ensure_stage_tables(jdsp_mfcc *h)
{
    if (h->stage_blob) return JDSP_OK;
    jdsp_ctx *ctx = h->ctx;
    const int NB = h->cfg.n_fft / 2, C = h->cfg.n_chan, NC = h->cfg.n_cep;
    const double PI = 3.141592;                                                  // MFCC:26
    std::vector<double> cosv((size_t)NC * C), lift(NC);
    for (int i = 1; i <= NC; i++) {
        for (int k = 1; k <= C; k++) cosv[(size_t)(i - 1) * C + (k - 1)] = cos(PI * i * (k - 0.5) / (double)C);
        lift[i - 1] = (1 + 0.5 * h->cfg.lifter * sin(PI * i / h->cfg.lifter));
    }
    const size_t o_fb = 0, o_cos = o_fb + sizeof(double) * NB, o_lift = o_cos + sizeof(double) * cosv.size(),
                 o_fi = o_lift + sizeof(double) * NC, total = o_fi + sizeof(int) * NB;
    std::vector<char> host(total);
    memcpy(&host[o_fb], h->fbank.data(), sizeof(double) * NB);
    memcpy(&host[o_cos], cosv.data(), sizeof(double) * cosv.size());
    memcpy(&host[o_lift], lift.data(), sizeof(double) * NC);
    memcpy(&host[o_fi], h->fi_bins.data(), sizeof(int) * NB);
    JDSP_HIP(ctx, hipMalloc(&h->stage_blob, total));
    JDSP_HIP(ctx, hipMemcpy(h->stage_blob, host.data(), total, hipMemcpyHostToDevice));
    char *b = (char *)h->stage_blob;
    h->stage_fb = (const double *)(b + o_fb);
    h->stage_cos = (const double *)(b + o_cos);
    h->stage_lift = (const double *)(b + o_lift);
    h->stage_fi = (const int *)(b + o_fi);
    return JDSP_OK;
}

}  // namespace

extern "C" {

int jdsp_mfcc_melfilterbank(jdsp_mfcc *h, const double *abs_host, long n_rows, double *mel_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_rows < 0 || n_rows > 0x7fffffffL) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_melfilterbank: n_rows");
    if (n_rows == 0) return JDSP_OK;
    if (!abs_host || !mel_host) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_melfilterbank: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage_tables(h);
    if (rc) return rc;
    const int NB = h->cfg.n_fft / 2, C = h->cfg.n_chan;
    Staged s(ctx);
    if ((rc = s.up(0, abs_host, sizeof(double) * (size_t)n_rows * NB)) || (rc = s.up(1, nullptr, sizeof(double) * (size_t)n_rows * C)))
        return rc;
    hipLaunchKernelGGL(jdsp::mel_filterbank_f64_kernel, dim3((unsigned)n_rows), dim3(64), 0, ctx->stream, (const double *)s.d[0],
                       n_rows, NB, C, h->stage_fi, h->stage_fb, (double *)s.d[1]);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return fail(ctx, JDSP_EHIP, "mel filterbank launch", le);
    return s.down(1, mel_host, sizeof(double) * (size_t)n_rows * C);
}

int jdsp_mfcc_dct(jdsp_mfcc *h, const double *mel_host, long n_rows, double *cep_inout_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_rows < 0 || n_rows > 0x7fffffffL) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_dct: n_rows");
    if (n_rows == 0) return JDSP_OK;
    if (!mel_host || !cep_inout_host) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_dct: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage_tables(h);
    if (rc) return rc;
    const int C = h->cfg.n_chan, NC = h->cfg.n_cep;
    Staged s(ctx);
    if ((rc = s.up(0, mel_host, sizeof(double) * (size_t)n_rows * C)) ||
        (rc = s.up(1, cep_inout_host, sizeof(double) * (size_t)n_rows * NC)))
        return rc;
    hipLaunchKernelGGL(jdsp::dct_f64_kernel, dim3((unsigned)n_rows), dim3(32), 0, ctx->stream, (const double *)s.d[0], n_rows, C,
                       NC, sqrt(2.0 / C), h->stage_cos, (double *)s.d[1]);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return fail(ctx, JDSP_EHIP, "dct launch", le);
    return s.down(1, cep_inout_host, sizeof(double) * (size_t)n_rows * NC);
}

int jdsp_mfcc_liftering(jdsp_mfcc *h, double *cep_inout_host, long n_rows)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_rows < 0 || n_rows > 0x7fffffffL) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_liftering: n_rows");
    if (n_rows == 0) return JDSP_OK;
    if (!cep_inout_host) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_liftering: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = ensure_stage_tables(h);
    if (rc) return rc;
    const int NC = h->cfg.n_cep;
    Staged s(ctx);
    if ((rc = s.up(0, cep_inout_host, sizeof(double) * (size_t)n_rows * NC))) return rc;
    hipLaunchKernelGGL(jdsp::lifter_f64_kernel, dim3((unsigned)n_rows), dim3(32), 0, ctx->stream, (double *)s.d[0], n_rows, NC,
                       h->stage_lift);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return fail(ctx, JDSP_EHIP, "lifter launch", le);
    return s.down(0, cep_inout_host, sizeof(double) * (size_t)n_rows * NC);
}

int jdsp_gmm_probability(jdsp_ctx *ctx, const double *feats_host, long n, const double *mean12, const double *cov144,
                         const double *eig48, double *prob_host)
{
    if (!ctx) return JDSP_EINVAL;
    if (n < 0) return fail(ctx, JDSP_EINVAL, "jdsp_gmm_probability: n < 0");
    if (n == 0) return JDSP_OK;
    if (!feats_host || !mean12 || !cov144 || !eig48 || !prob_host) return fail(ctx, JDSP_EINVAL, "jdsp_gmm_probability: NULL buffer");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    // what probability() reads of its arguments: mean[i], covariance[i][i], i < 4 (GMMTest:232), all of eigenVector;
    // the normalisation (1/sqrt(2 PI)) (1/sqrt(c)) in the reference's order, PI 3.141592 (GMMTest:19)
    double par[12 + 48];
    for (int i = 0; i < 4; i++) {
        par[i] = mean12[i];
        par[4 + i] = cov144[12 * i + i];
        par[8 + i] = (1.0 / sqrt(2.0 * 3.141592)) * (1.0 / sqrt(cov144[12 * i + i]));
    }
    memcpy(par + 12, eig48, sizeof(double) * 48);
    Staged s(ctx);
    int rc;
    if ((rc = s.up(0, feats_host, sizeof(double) * 12 * (size_t)n)) || (rc = s.up(1, par, sizeof(par))) ||
        (rc = s.up(2, nullptr, sizeof(double) * (size_t)n)))
        return rc;
    hipLaunchKernelGGL(jdsp::gmm_probability_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       (const double *)s.d[0], n, (const double *)s.d[1], (double *)s.d[2]);
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return fail(ctx, JDSP_EHIP, "gmm probability launch", le);
    return s.down(2, prob_host, sizeof(double) * (size_t)n);
}

}  // extern "C"
