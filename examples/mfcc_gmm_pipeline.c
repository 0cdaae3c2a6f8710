/* mfcc_gmm_pipeline.c -- the C ABI from plain C, no HIP headers on the caller's side:
 * int16 PCM -> MFCC vectors -> GMM class scores, with the vectors never leaving HBM
 * (MFCCFeatureExtraction_auto_version1.cpp feeding GMMAlgorithm_Test_Auto_ver2.cpp without the .mfc files).
 *
 *   cc -std=c99 -Iinclude examples/mfcc_gmm_pipeline.c -Ljeicyboodsp_amd -ljdsp -Wl,-rpath,$PWD/jeicyboodsp_amd -o pipeline
 *   ./pipeline pcm.raw params.bin n_classes utt_len_blocks  >  scores.txt
 *
 * pcm.raw: int16 mono; params.bin: n_classes GMMParameter records (GMMTest:29-34); every utt_len_blocks
 * blocks of 512 samples form one utterance (frames never straddle utterances).  Prints, per utterance, the
 * arg-max class (1-based, as GMMTest:127 prints it) and the class scores. */
#include <stdio.h>
#include <stdlib.h>

#include "jdsp.h"

#define CK(call)                                                                   \
    do {                                                                           \
        if ((call) != JDSP_OK) {                                                   \
            fprintf(stderr, "%s: %s\n", #call, jdsp_last_error(ctx));              \
            return 2;                                                              \
        }                                                                          \
    } while (0)

static void *slurp(const char *path, size_t *bytes)
{
    FILE *f = fopen(path, "rb");
    void *p;
    long n;
    if (!f) { perror(path); exit(1); }
    fseek(f, 0L, SEEK_END);
    n = ftell(f);
    fseek(f, 0L, SEEK_SET);
    p = malloc(n > 0 ? (size_t)n : 1);
    if (n > 0 && fread(p, 1, (size_t)n, f) != (size_t)n) { perror(path); exit(1); }
    fclose(f);
    *bytes = (size_t)n;
    return p;
}

int main(int argc, char **argv)
{
    jdsp_ctx *ctx = NULL;
    jdsp_mfcc *mf = NULL;
    jdsp_gmm *gm = NULL;
    jdsp_mfcc_cfg cfg;
    size_t pcm_bytes, par_bytes;
    int16_t *pcm;
    jdsp_gmm_param *par;
    int n_classes, utt_blocks;
    long n_samples, n_utts, frames_per_utt, n_frames, u, k, c;
    int64_t *starts, *first;
    void *d_pcm, *d_starts, *d_feats, *d_first, *d_scores, *d_best;
    double *scores;
    int *best;

    if (argc != 5) { fprintf(stderr, "usage: %s pcm.raw params.bin n_classes utt_len_blocks\n", argv[0]); return 1; }
    pcm = (int16_t *)slurp(argv[1], &pcm_bytes);
    par = (jdsp_gmm_param *)slurp(argv[2], &par_bytes);
    n_classes = atoi(argv[3]);
    utt_blocks = atoi(argv[4]);
    if (n_classes < 1 || (size_t)n_classes * sizeof(jdsp_gmm_param) > par_bytes || utt_blocks < 2) return 1;
    n_samples = (long)(pcm_bytes / 2);
    n_utts = n_samples / (512L * utt_blocks);
    frames_per_utt = utt_blocks - 1;                      /* 1024-sample frames at hop 512 inside one utterance */
    n_frames = n_utts * frames_per_utt;
    if (n_utts < 1) return 1;

    if (jdsp_create(0, &ctx) != JDSP_OK) { fprintf(stderr, "jdsp_create: %s\n", jdsp_last_error(NULL)); return 2; }
    CK(jdsp_mfcc_native_cfg(&cfg));
    CK(jdsp_mfcc_create(ctx, &cfg, &mf));
    CK(jdsp_gmm_create(ctx, par, n_classes, &gm));

    /* frame j of utterance u starts at sample 512 (u * utt_blocks + j); utterance u owns vectors [u F, (u+1) F) */
    starts = (int64_t *)malloc(sizeof(int64_t) * (size_t)n_frames);
    first = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_utts + 1));
    for (u = 0; u < n_utts; u++)
        for (k = 0; k < frames_per_utt; k++) starts[u * frames_per_utt + k] = 512 * (u * utt_blocks + k);
    for (u = 0; u <= n_utts; u++) first[u] = u * frames_per_utt;

    CK(jdsp_malloc(ctx, (size_t)n_samples * 2, &d_pcm));
    CK(jdsp_malloc(ctx, sizeof(int64_t) * (size_t)n_frames, &d_starts));
    CK(jdsp_malloc(ctx, sizeof(double) * 12 * (size_t)n_frames, &d_feats));
    CK(jdsp_malloc(ctx, sizeof(int64_t) * (size_t)(n_utts + 1), &d_first));
    CK(jdsp_malloc(ctx, sizeof(double) * (size_t)n_utts * (size_t)n_classes, &d_scores));
    CK(jdsp_malloc(ctx, sizeof(int) * (size_t)n_utts, &d_best));
    CK(jdsp_memcpy_h2d(ctx, d_pcm, pcm, (size_t)n_samples * 2));
    CK(jdsp_memcpy_h2d(ctx, d_starts, starts, sizeof(int64_t) * (size_t)n_frames));
    CK(jdsp_memcpy_h2d(ctx, d_first, first, sizeof(int64_t) * (size_t)(n_utts + 1)));

    /* two enqueues on the handle's stream; the vectors stay in d_feats */
    CK(jdsp_mfcc_frames_dev(mf, (const int16_t *)d_pcm, (const int64_t *)d_starts, n_frames, (double *)d_feats));
    CK(jdsp_gmm_score_dev(gm, (const double *)d_feats, n_frames, (const int64_t *)d_first, n_utts, (double *)d_scores,
                          (int *)d_best));

    scores = (double *)malloc(sizeof(double) * (size_t)n_utts * (size_t)n_classes);
    best = (int *)malloc(sizeof(int) * (size_t)n_utts);
    CK(jdsp_memcpy_d2h(ctx, scores, d_scores, sizeof(double) * (size_t)n_utts * (size_t)n_classes));   /* synchronises */
    CK(jdsp_memcpy_d2h(ctx, best, d_best, sizeof(int) * (size_t)n_utts));
    for (u = 0; u < n_utts; u++) {
        printf("%ld %d", u, best[u] + 1);
        for (c = 0; c < n_classes; c++) printf(" %.17g", scores[u * n_classes + c]);
        printf("\n");
    }

    jdsp_free(ctx, d_pcm); jdsp_free(ctx, d_starts); jdsp_free(ctx, d_feats);
    jdsp_free(ctx, d_first); jdsp_free(ctx, d_scores); jdsp_free(ctx, d_best);
    jdsp_gmm_destroy(gm);
    jdsp_mfcc_destroy(mf);
    jdsp_destroy(ctx);
    free(pcm); free(par); free(starts); free(first); free(scores); free(best);
    return 0;
}
