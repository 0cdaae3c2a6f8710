// gmm_api.hip -- C ABI of GMM scoring (GMMAlgorithm_Test_Auto_ver2.cpp) and the HMM recursion
// (Viterbi_version1.cpp) on MFCC vectors that are already in HBM.
#include "jdsp_internal.h"

using jdsp::fail;

namespace {

// What probability() reads of a GMMParameter (GMMTest:216-235): eigenVector[k], mean[k][0..3] and the first
// four diagonal entries of covariance[k]; the normalisation (1/sqrt(2 PI)) (1/sqrt(c)) of :232 is evaluated
// here with the host's sqrt, in the reference's order.
void pack_gmm(const jdsp_gmm_param &p, double *r)
{
    const double PI = 3.141592;                                                  // GMMTest:23
    for (int k = 0; k < 4; k++) {
        r[jdsp::kGmmAlpa + k] = p.alpa[k];
        for (int i = 0; i < 4; i++) {
            const double c = p.covariance[k][i][i];
            r[jdsp::kGmmMean + 4 * k + i] = p.mean[k][i];
            r[jdsp::kGmmVar + 4 * k + i] = c;
            r[jdsp::kGmmCoef + 4 * k + i] = (1.0 / sqrt(2.0 * PI)) * (1.0 / sqrt(c));
            r[jdsp::kGmmNhiv + 4 * k + i] = -0.5 / c;
        }
        double cp = 1.0;
        for (int i = 0; i < 4; i++) cp *= r[jdsp::kGmmCoef + 4 * k + i];
        r[jdsp::kGmmCprod + k] = cp;
        for (int i = 0; i < 12; i++)
            for (int j = 0; j < 4; j++) r[jdsp::kGmmEig + 48 * k + 4 * i + j] = p.eigenVector[k][i][j];
    }
}

int check_batch(jdsp_ctx *ctx, const char *who, const void *feats, const void *utt_first, long n_utts, long n_frames)
{
    if (n_utts < 0 || n_frames < 0) return fail(ctx, JDSP_EINVAL, who);
    if (n_utts > 0 && !utt_first) return fail(ctx, JDSP_EINVAL, who);
    if (n_frames > 0 && (!feats || ((uintptr_t)feats & 15u))) return fail(ctx, JDSP_EINVAL, who);
    return JDSP_OK;
}

// host-side validation of the utterance table of the host entries: first[0] >= 0, non-decreasing
long check_offsets(const int64_t *first, long n_utts)
{
    long prev = 0;
    for (long u = 0; u <= n_utts; u++) {
        if (first[u] < prev) return -1;
        prev = (long)first[u];
    }
    return prev;
}

}  // namespace

extern "C" {

int jdsp_gmm_create(jdsp_ctx *ctx, const jdsp_gmm_param *classes, int n_classes, jdsp_gmm **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    if (!classes || n_classes < 1 || n_classes > jdsp::kGmmMaxClasses)
        return fail(ctx, JDSP_EINVAL, "jdsp_gmm_create: 1..256 classes");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    jdsp_gmm *h = new (std::nothrow) jdsp_gmm();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_gmm_create");
    h->ctx = ctx;
    h->n_classes = n_classes;
    std::vector<double> rec((size_t)n_classes * jdsp::kGmmRecord);
    for (int c = 0; c < n_classes; c++) pack_gmm(classes[c], &rec[(size_t)c * jdsp::kGmmRecord]);
    hipError_t e = hipMalloc(&h->records, rec.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->records, rec.data(), rec.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jdsp_gmm_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_gmm_create: upload", e);
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_gmm_destroy(jdsp_gmm *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->records) (void)hipFree(h->records);
    delete h;
    return JDSP_OK;
}

int jdsp_gmm_set_option(jdsp_gmm *h, const char *name, long value)
{
    if (!h || !name) return JDSP_EINVAL;
    if (!strcmp(name, "evaluation")) {
        if (value != 0 && value != 1) return fail(h->ctx, JDSP_EINVAL, "jdsp_gmm_set_option: evaluation is 0 or 1");
        h->fused = (int)value;
        return JDSP_OK;
    }
    return fail(h->ctx, JDSP_EINVAL, "jdsp_gmm_set_option: unknown option");
}

int jdsp_gmm_score_dev(jdsp_gmm *h, const double *feats_dev, long n_frames, const int64_t *utt_first_dev, long n_utts,
                       double *scores_dev, int *best_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    int rc = check_batch(ctx, "jdsp_gmm_score_dev: bad buffer", feats_dev, utt_first_dev, n_utts, n_frames);
    if (rc) return rc;
    if (n_utts > 0 && !scores_dev) return fail(ctx, JDSP_EINVAL, "jdsp_gmm_score_dev: scores required");
    if (n_utts == 0) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (jdsp::launch_gmm_score(ctx->stream, feats_dev, n_frames, (const long long *)utt_first_dev, n_utts, h->records, h->n_classes,
                               h->fused, scores_dev, best_dev))
        return fail(ctx, JDSP_EHIP, "gmm score launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_gmm_score(jdsp_gmm *h, const double *feats_host, const int64_t *utt_first_host, long n_utts,
                   double *scores_host, int *best_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_utts < 0 || (n_utts > 0 && (!utt_first_host || !scores_host)))
        return fail(ctx, JDSP_EINVAL, "jdsp_gmm_score: bad buffer");
    if (n_utts == 0) return JDSP_OK;
    const long n_frames = check_offsets(utt_first_host, n_utts);
    if (n_frames < 0 || utt_first_host[0] != 0 || (n_frames > 0 && !feats_host))
        return fail(ctx, JDSP_EINVAL, "jdsp_gmm_score: bad utterance table");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    double *d_feats = nullptr, *d_scores = nullptr;
    int64_t *d_first = nullptr;
    int *d_best = nullptr;
    const size_t sz_scores = (size_t)n_utts * h->n_classes * sizeof(double);
    hipError_t e = hipMalloc(&d_feats, (size_t)(n_frames > 0 ? n_frames : 1) * 12 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_first, (size_t)(n_utts + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&d_scores, sz_scores);
    if (e == hipSuccess) e = hipMalloc(&d_best, (size_t)n_utts * sizeof(int));
    if (e == hipSuccess && n_frames > 0)
        e = hipMemcpyAsync(d_feats, feats_host, (size_t)n_frames * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_first, utt_first_host, (size_t)(n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    int rc = JDSP_OK;
    if (e == hipSuccess) rc = jdsp_gmm_score_dev(h, d_feats, n_frames, d_first, n_utts, d_scores, d_best);
    if (e == hipSuccess && rc == JDSP_OK) e = hipMemcpyAsync(scores_host, d_scores, sz_scores, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == JDSP_OK && best_host)
        e = hipMemcpyAsync(best_host, d_best, (size_t)n_utts * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    else (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_feats); (void)hipFree(d_first); (void)hipFree(d_scores); (void)hipFree(d_best);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, JDSP_EHIP, "jdsp_gmm_score", e);
    return JDSP_OK;
}

int jdsp_hmm_create(jdsp_ctx *ctx, const jdsp_hmm_param *models, int n_models, jdsp_hmm **out)
{
    if (!ctx || !out) return JDSP_EINVAL;
    *out = nullptr;
    if (!models || n_models < 1 || n_models > 1024) return fail(ctx, JDSP_EINVAL, "jdsp_hmm_create: 1..1024 models");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    jdsp_hmm *h = new (std::nothrow) jdsp_hmm();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_hmm_create");
    h->ctx = ctx;
    h->n_models = n_models;
    std::vector<double> rec((size_t)n_models * 6 * jdsp::kGmmRecord), lt((size_t)n_models * 36);
    for (int m = 0; m < n_models; m++) {
        for (int s = 0; s < 6; s++) pack_gmm(models[m].gMMParam[s], &rec[((size_t)m * 6 + s) * jdsp::kGmmRecord]);
        for (int u = 0; u < 6; u++)
            for (int v = 0; v < 6; v++) lt[(size_t)m * 36 + 6 * u + v] = log(models[m].transProb[u][v]);   // Viterbi:196
    }
    hipError_t e = hipMalloc(&h->records, rec.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&h->log_trans, lt.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(h->records, rec.data(), rec.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(h->log_trans, lt.data(), lt.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jdsp_hmm_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_hmm_create: upload", e);
    }
    *out = h;
    return JDSP_OK;
}

int jdsp_hmm_destroy(jdsp_hmm *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->records) (void)hipFree(h->records);
    if (h->log_trans) (void)hipFree(h->log_trans);
    if (h->emission) (void)hipFree(h->emission);
    delete h;
    return JDSP_OK;
}

int jdsp_hmm_set_option(jdsp_hmm *h, const char *name, long value)
{
    if (!h || !name) return JDSP_EINVAL;
    if (!strcmp(name, "evaluation")) {
        if (value != 0 && value != 1) return fail(h->ctx, JDSP_EINVAL, "jdsp_hmm_set_option: evaluation is 0 or 1");
        h->fused = (int)value;
        return JDSP_OK;
    }
    return fail(h->ctx, JDSP_EINVAL, "jdsp_hmm_set_option: unknown option");
}

int jdsp_hmm_reserve(jdsp_hmm *h, long n_frames)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_frames <= h->emission_frames) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h->emission) (void)hipFree(h->emission);
    h->emission = nullptr;
    h->emission_frames = 0;
    JDSP_HIP(ctx, hipMalloc(&h->emission, (size_t)n_frames * h->n_models * 6 * sizeof(double)));
    h->emission_frames = n_frames;
    return JDSP_OK;
}

int jdsp_hmm_viterbi_dev(jdsp_hmm *h, const double *feats_dev, long n_frames, const int64_t *utt_first_dev, long n_utts,
                         double *scores_dev, int *best_dev, int *path_dev, double *trellis_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    int rc = check_batch(ctx, "jdsp_hmm_viterbi_dev: bad buffer", feats_dev, utt_first_dev, n_utts, n_frames);
    if (rc) return rc;
    if (n_utts == 0) return JDSP_OK;
    rc = jdsp_hmm_reserve(h, n_frames > 0 ? n_frames : 1);
    if (rc) return rc;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    const double log_init = log(1.0 / 6.0);                                      // Viterbi:186
    if (jdsp::launch_hmm_viterbi(ctx->stream, feats_dev, n_frames, (const long long *)utt_first_dev, n_utts, h->records,
                                 h->log_trans, h->n_models, h->fused, log_init, h->emission, scores_dev, best_dev, path_dev,
                                 trellis_dev))
        return fail(ctx, JDSP_EHIP, "hmm launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_hmm_viterbi(jdsp_hmm *h, const double *feats_host, const int64_t *utt_first_host, long n_utts,
                     double *scores_host, int *best_host, int *path_host, double *trellis_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_utts < 0 || (n_utts > 0 && !utt_first_host)) return fail(ctx, JDSP_EINVAL, "jdsp_hmm_viterbi: bad buffer");
    if (n_utts == 0) return JDSP_OK;
    const long n_frames = check_offsets(utt_first_host, n_utts);
    if (n_frames < 0 || utt_first_host[0] != 0 || (n_frames > 0 && !feats_host))
        return fail(ctx, JDSP_EINVAL, "jdsp_hmm_viterbi: bad utterance table");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    double *d_feats = nullptr, *d_scores = nullptr;
    int64_t *d_first = nullptr;
    int *d_best = nullptr, *d_path = nullptr;
    double *d_trellis = nullptr;
    const size_t sz_trellis = (size_t)h->n_models * 6 * (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(double);
    const size_t sz_scores = (size_t)n_utts * h->n_models * sizeof(double);
    const size_t sz_path = (size_t)h->n_models * (size_t)(n_frames > 0 ? n_frames : 1) * sizeof(int);
    hipError_t e = hipMalloc(&d_feats, (size_t)(n_frames > 0 ? n_frames : 1) * 12 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&d_first, (size_t)(n_utts + 1) * sizeof(int64_t));
    if (e == hipSuccess) e = hipMalloc(&d_scores, sz_scores);
    if (e == hipSuccess) e = hipMalloc(&d_best, (size_t)n_utts * sizeof(int));
    if (e == hipSuccess && path_host) e = hipMalloc(&d_path, sz_path);
    if (e == hipSuccess && trellis_host) e = hipMalloc(&d_trellis, sz_trellis);
    if (e == hipSuccess && n_frames > 0)
        e = hipMemcpyAsync(d_feats, feats_host, (size_t)n_frames * 12 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(d_first, utt_first_host, (size_t)(n_utts + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream);
    int rc = JDSP_OK;
    if (e == hipSuccess) rc = jdsp_hmm_viterbi_dev(h, d_feats, n_frames, d_first, n_utts, d_scores, d_best, d_path, d_trellis);
    if (e == hipSuccess && rc == JDSP_OK && scores_host)
        e = hipMemcpyAsync(scores_host, d_scores, sz_scores, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == JDSP_OK && best_host)
        e = hipMemcpyAsync(best_host, d_best, (size_t)n_utts * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == JDSP_OK && path_host && n_frames > 0)
        e = hipMemcpyAsync(path_host, d_path, (size_t)h->n_models * n_frames * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == JDSP_OK && trellis_host && n_frames > 0)
        e = hipMemcpyAsync(trellis_host, d_trellis, (size_t)h->n_models * 6 * n_frames * sizeof(double), hipMemcpyDeviceToHost,
                           ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    else (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_feats); (void)hipFree(d_first); (void)hipFree(d_scores); (void)hipFree(d_best); (void)hipFree(d_path);
    (void)hipFree(d_trellis);
    if (rc) return rc;
    if (e != hipSuccess) return fail(ctx, JDSP_EHIP, "jdsp_hmm_viterbi", e);
    return JDSP_OK;
}

}  // extern "C"
