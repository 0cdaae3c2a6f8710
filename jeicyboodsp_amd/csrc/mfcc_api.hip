// mfcc_api.hip -- C ABI of the MFCC front end; MelFilterBankInit runs here, on the host.
#include "jdsp_internal.h"

using jdsp::fail;

extern "C" {

int jdsp_mfcc_native_cfg(jdsp_mfcc_cfg *c)
{
    if (!c) return JDSP_EINVAL;
    // MFCCFeatureExtraction_auto_version1.cpp:23-33
    c->win_len = 1024; c->hop = 512; c->n_fft = 1024; c->n_chan = 38; c->n_cep = 12; c->lifter = 22;
    c->half_rate = 22050.0; c->preemph = 0.96;
    return JDSP_OK;
}

int jdsp_mfcc_create(jdsp_ctx *ctx, const jdsp_mfcc_cfg *cfg, jdsp_mfcc **out)
{
    if (!ctx || !cfg || !out) return JDSP_EINVAL;
    *out = nullptr;
    const jdsp_mfcc_cfg c = *cfg;
    if ((c.n_fft != 1024 && c.n_fft != 512) || c.win_len < 2 || c.win_len > c.n_fft || c.hop < 1 || c.n_chan < 1 ||
        c.n_chan > 64 || c.n_cep < 1 || c.n_cep > 32 || c.lifter < 1 || !(c.half_rate > 0))
        return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_create: unsupported configuration");
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int rc = jdsp::ensure_stft1024_table(ctx);
    if (rc) return rc;
    jdsp_mfcc *h = new (std::nothrow) jdsp_mfcc();
    if (!h) return fail(ctx, JDSP_ENOMEM, "jdsp_mfcc_create");
    h->ctx = ctx;
    h->cfg = c;
    const int C = c.n_chan, NB = c.n_fft / 2;
    const double PI = 3.141592;                                                  // :26
    // ---- MelFilterBankInit (:118-152), in double like the reference ----
    h->mel_freqs.assign(C + 1, 0.0);
    h->fi_bins.assign(NB, 0);
    h->fbank.assign(NB, 0.0);
    const double unit = 1127.0 * log(1 + (c.half_rate / 700.0)) / (C + 1);       // :124
    for (int i = 1; i <= C + 1; i++) h->mel_freqs[i - 1] = 700 * (exp(unit * i / 1127.0) - 1.0);   // :126-129
    for (int i = 0, k = 0; i < NB; i++) {                                        // :131-137
        if ((i / (double)(NB - 1)) * c.half_rate > h->mel_freqs[k])
            if (k < C) k++;
        h->fi_bins[i] = k;
    }
    for (int i = 0; i < NB; i++) {                                               // :139-150
        const int k = h->fi_bins[i];
        const double fr = (i / (double)(NB - 1)) * c.half_rate;
        double v = k == 0 ? (h->mel_freqs[0] - fr) / (h->mel_freqs[0] - 0)
                          : (h->mel_freqs[k] - fr) / (h->mel_freqs[k] - h->mel_freqs[k - 1]);
        h->fbank[i] = v < 0 ? 0 : v;
    }
    // ---- per-bin filterbank weight and channel index for the kernel (MelFilterBank, :157-168) ----
    std::vector<float> mel_fb(512, 0.f);
    std::vector<int> mel_k(512, 0);
    for (int i = 0; i < 512; i++) {
        mel_fb[i] = i < NB ? (float)h->fbank[i] : 0.f;
        mel_k[i] = i < NB ? h->fi_bins[i] : h->fi_bins[NB - 1];
    }
    // ---- the same filterbank by channel index: every run of bins with one rgdFiBins value, cut into pieces of
    // at most piece_len bins, one piece per lane (the index never decreases with the bin, :131-137).  piece_len is
    // the smallest of 8, 12, 16 for which the pieces fit 64 lanes with at most four per channel: the kernels walk
    // piece_len bins on every lane, so shorter pieces are less work (native 512 bins / 38 channels: 12; 256 / 40: 8) ----
    std::vector<int> seg, chan_src;
    std::vector<float> seg_wc;
    int seg_lanes = 0, piece_len = 16;
    bool seg_ok = true, chan_ok = true;
    for (int want : {8, 12, 16}) {
        piece_len = want;
        seg.assign(64 * 4, 0);
        seg_wc.assign(2 * 16 * 64, 0.f);
        seg_lanes = 0;
        seg_ok = true;
        for (int i = 0; i < NB && seg_ok;) {
            int e = i;
            while (e < NB && h->fi_bins[e] == h->fi_bins[i]) e++;
            for (int s0 = i; s0 < e; s0 += piece_len) {
                if (seg_lanes == 64) { seg_ok = false; break; }
                const int cnt = e - s0 < piece_len ? e - s0 : piece_len;
                seg[4 * seg_lanes + 0] = s0; seg[4 * seg_lanes + 1] = cnt; seg[4 * seg_lanes + 2] = h->fi_bins[i];
                for (int t = 0; t < cnt; t++) {
                    const size_t at = ((size_t)(t >> 1) * 64 + seg_lanes) * 4 + 2 * (t & 1);       // MfccDev::seg_wc
                    seg_wc[at] = (float)h->fbank[s0 + t];
                    seg_wc[at + 1] = (float)(1.0 - h->fbank[s0 + t]);                             // (1 - rgdFilterBank[i]), :161,:165
                }
                seg_lanes++;
            }
            i = e;
        }
        // ---- per channel: which lanes' pieces feed it (mel_channel_sums) ----
        chan_src.assign(64 * 4, 0);
        chan_ok = seg_ok;
        for (int ch = 0; ch < C && chan_ok; ch++) {
            int h0 = -1, hc = 0, l0 = -1, lc = 0;
            for (int L = 0; L < seg_lanes; L++) {
                const int z = seg[4 * L + 2];
                if (z == ch) { if (h0 < 0) h0 = L; hc++; }
                if (z == ch + 1) { if (l0 < 0) l0 = L; lc++; }
            }
            if (hc > 4 || lc > 4) chan_ok = false;
            chan_src[4 * ch + 0] = h0 < 0 ? 0 : h0; chan_src[4 * ch + 1] = hc;
            chan_src[4 * ch + 2] = l0 < 0 ? 0 : l0; chan_src[4 * ch + 3] = lc;
        }
        if (seg_ok && chan_ok) break;
    }
    // ---- DCT (:178-182) and lifter (:189) constants ----
    std::vector<double> dct((size_t)(C > 40 ? C : 40) * 32, 0.0), lift(32, 0.0);   // zero rows up to 40: mfcc_x2_kernel reads ten per lane group unguarded
    for (int i = 1; i <= c.n_cep; i++) {
        for (int k = 1; k <= C; k++) dct[(size_t)(k - 1) * 32 + (i - 1)] = sqrt(2.0 / C) * cos(PI * i * (k - 0.5) / (double)C);
        lift[i - 1] = (1 + 0.5 * c.lifter * sin(PI * i / c.lifter));
    }
    // ---- Hamming over win_len (:213), halved for the split, zero beyond ----
    std::vector<float2> window(512, make_float2(0.f, 0.f));
    for (int i = 0; i < c.win_len; i++) {
        const double w = 0.5 * (0.54 - 0.46 * cos(2 * PI * i / (c.win_len - 1)));
        if (i & 1) window[i >> 1].y = (float)w;
        else window[i >> 1].x = (float)w;
    }
    // ---- one blob ----
    const size_t o_win = 0, o_fb = o_win + sizeof(float2) * 512, o_k = o_fb + 512 * sizeof(float),
                 o_dct = o_k + 512 * sizeof(int), o_lift = o_dct + dct.size() * sizeof(double),
                 o_seg = o_lift + 32 * sizeof(double), o_segw = o_seg + seg.size() * sizeof(int),
                 o_chan = o_segw + seg_wc.size() * sizeof(float),
                 total = o_chan + chan_src.size() * sizeof(int);
    std::vector<char> host(total, 0);
    memcpy(&host[o_win], window.data(), sizeof(float2) * 512);
    memcpy(&host[o_fb], mel_fb.data(), 512 * sizeof(float));
    memcpy(&host[o_k], mel_k.data(), 512 * sizeof(int));
    memcpy(&host[o_dct], dct.data(), dct.size() * sizeof(double));
    memcpy(&host[o_lift], lift.data(), 32 * sizeof(double));
    memcpy(&host[o_seg], seg.data(), seg.size() * sizeof(int));
    memcpy(&host[o_segw], seg_wc.data(), seg_wc.size() * sizeof(float));
    memcpy(&host[o_chan], chan_src.data(), chan_src.size() * sizeof(int));
    hipError_t e = hipMalloc(&h->blob, total);
    if (e == hipSuccess) e = hipMemcpy(h->blob, host.data(), total, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        jdsp_mfcc_destroy(h);
        return fail(ctx, JDSP_EHIP, "jdsp_mfcc_create: tables", e);
    }
    char *b = (char *)h->blob;
    h->dev.win_len = c.win_len; h->dev.hop = c.hop; h->dev.n_chan = C; h->dev.n_cep = c.n_cep;
    h->dev.bin_stride = c.n_fft == 512 ? 2 : 1;
    h->dev.preemph = (float)c.preemph;
    h->dev.window = (const float2 *)(b + o_win);
    h->dev.n_bins = NB;
    h->dev.mel_fb = (const float *)(b + o_fb);
    h->dev.mel_k = (const int *)(b + o_k);
    h->dev.seg = (const int4 *)(b + o_seg);
    h->dev.seg_wc = (const float4 *)(b + o_segw);
    h->dev.piece_len = piece_len;
    h->dev.seg_ok = seg_ok ? 1 : 0;
    h->dev.chan_src = (const int4 *)(b + o_chan);
    h->dev.chan_ok = chan_ok ? 1 : 0;
    h->dev.dct = (const double *)(b + o_dct);
    h->dev.lifter_w = (const double *)(b + o_lift);
    *out = h;
    return JDSP_OK;
}

int jdsp_mfcc_destroy(jdsp_mfcc *h)
{
    if (!h) return JDSP_OK;
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (h->blob) (void)hipFree(h->blob);
    if (h->redo) (void)hipFree(h->redo);
    if (h->stage_blob) (void)hipFree(h->stage_blob);
    delete h;
    return JDSP_OK;
}

int jdsp_mfcc_tables(const jdsp_mfcc *h, double *mel_freqs, int *fi_bins, double *fbank)
{
    if (!h) return JDSP_EINVAL;
    if (mel_freqs) memcpy(mel_freqs, h->mel_freqs.data(), h->mel_freqs.size() * sizeof(double));
    if (fi_bins) memcpy(fi_bins, h->fi_bins.data(), h->fi_bins.size() * sizeof(int));
    if (fbank) memcpy(fbank, h->fbank.data(), h->fbank.size() * sizeof(double));
    return JDSP_OK;
}

int jdsp_mfcc_frames_dev(jdsp_mfcc *h, const int16_t *pcm_dev, const int64_t *frame_start_dev, long n_frames,
                         double *feats_dev)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_frames < 0 || (n_frames > 0 && (!pcm_dev || !feats_dev))) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_frames: bad buffer");
    if (n_frames == 0) return JDSP_OK;
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    if (h->dev.bin_stride == 2 && (n_frames + 1) / 2 > h->redo_cap) {        // no-op once sized (size it before a graph capture)
        JDSP_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (h->redo) (void)hipFree(h->redo);
        h->redo = nullptr;
        h->redo_cap = 0;
        JDSP_HIP(ctx, hipMalloc((void **)&h->redo, ((size_t)(n_frames + 1) / 2 + 1) * sizeof(int)));
        h->redo_cap = (n_frames + 1) / 2;
    }
    if (jdsp::launch_mfcc(ctx->stream, pcm_dev, (const long long *)frame_start_dev, n_frames, h->dev, ctx->stft1024_table,
                          feats_dev, h->redo))
        return fail(ctx, JDSP_EHIP, "mfcc launch", hipGetLastError());
    return JDSP_OK;
}

int jdsp_mfcc_frames(jdsp_mfcc *h, const int16_t *pcm_host, long n_samples, const int64_t *frame_start_host,
                     long n_frames, double *feats_host)
{
    if (!h) return JDSP_EINVAL;
    jdsp_ctx *ctx = h->ctx;
    if (n_frames < 0 || n_samples < 0) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_frames: negative size");
    if (n_frames == 0) return JDSP_OK;
    if (!pcm_host || !feats_host) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_frames: NULL buffer");
    for (long j = 0; j < n_frames; j++) {          // every frame must lie inside the buffer
        const long long st = frame_start_host ? frame_start_host[j] : (long long)h->cfg.hop * j;
        if (st < 0 || st + h->cfg.win_len > n_samples) return fail(ctx, JDSP_EINVAL, "jdsp_mfcc_frames: frame outside pcm");
    }
    JDSP_HIP(ctx, hipSetDevice(ctx->device));
    int16_t *d_in = nullptr;
    int64_t *d_st = nullptr;
    double *d_out = nullptr;
    const size_t out_b = (size_t)n_frames * h->cfg.n_cep * sizeof(double);
    hipError_t e = hipMalloc((void **)&d_in, (size_t)n_samples * 2);
    if (e == hipSuccess) e = hipMalloc((void **)&d_out, out_b);
    if (e == hipSuccess && frame_start_host) e = hipMalloc((void **)&d_st, (size_t)n_frames * 8);
    int rc = JDSP_OK;
    if (e != hipSuccess) rc = fail(ctx, JDSP_ENOMEM, "jdsp_mfcc_frames: hipMalloc", e);
    if (!rc && (e = hipMemcpyAsync(d_in, pcm_host, (size_t)n_samples * 2, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mfcc_frames: H2D", e);
    if (!rc && frame_start_host &&
        (e = hipMemcpyAsync(d_st, frame_start_host, (size_t)n_frames * 8, hipMemcpyHostToDevice, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mfcc_frames: H2D", e);
    if (!rc) rc = jdsp_mfcc_frames_dev(h, d_in, d_st, n_frames, d_out);
    if (!rc && (e = hipMemcpyAsync(feats_host, d_out, out_b, hipMemcpyDeviceToHost, ctx->stream)) != hipSuccess)
        rc = fail(ctx, JDSP_EHIP, "jdsp_mfcc_frames: D2H", e);
    if ((e = hipStreamSynchronize(ctx->stream)) != hipSuccess && !rc) rc = fail(ctx, JDSP_EHIP, "jdsp_mfcc_frames: sync", e);
    if (d_in) (void)hipFree(d_in);
    if (d_st) (void)hipFree(d_st);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

}  // extern "C"
