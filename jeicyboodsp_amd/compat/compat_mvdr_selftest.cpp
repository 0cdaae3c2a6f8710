// compat_mvdr_selftest.cpp -- BeamForming_MVDR_ver1.cpp's main() loop (:83-109) in structure, on the per-block
// functions of jeicyboo_compat_mvdr.h.  usage: compat_mvdr_selftest left.raw right.raw out.bin [ooo]
// out.bin: the int16 blocks the loop writes, then the final rgdSpatialCorr (4 doubles).
// "vad" mode: VoiceActivityDetection (BF:207-242) on every block of left.raw, one byte per block.
#include <cstdio>
#include <cstring>

#include "jeicyboo_compat_mvdr.h"

#define BLOCK_LEN 512

// "ooo" mode: the two functions called OUT of main()'s order.  ProcessMVDR on blocks 0..2 with a hand-set matrix and
// no estimate at all; then EstimateSpatialCorrMtx on the frames [block 3, block 4] and [block 7, block 5] (not even
// consecutive blocks) into that pre-filled matrix; then ProcessMVDR on blocks 3..5 with the result.
static int out_of_order(FILE *l, FILE *r, FILE *w)
{
    static short L[8][BLOCK_LEN], R[8][BLOCK_LEN];
    for (int b = 0; b < 8; b++)
        if (fread(L[b], sizeof(short), BLOCK_LEN, l) != BLOCK_LEN || fread(R[b], sizeof(short), BLOCK_LEN, r) != BLOCK_LEN) return 1;
    double corr[2][2] = {{4.0e6, 1.5e5}, {-2.5e5, 3.0e6}};
    short out[BLOCK_LEN], tl[2 * BLOCK_LEN], tr[2 * BLOCK_LEN];
    for (int b = 0; b < 3; b++)
        if (ProcessMVDR(L[b], R[b], BLOCK_LEN, out, 0.0, corr)) fwrite(out, sizeof(short), BLOCK_LEN, w);
    const int pairs[2][2] = {{3, 4}, {7, 5}};
    for (auto &p : pairs) {
        memcpy(tl, L[p[0]], sizeof(L[0])); memcpy(tl + BLOCK_LEN, L[p[1]], sizeof(L[0]));
        memcpy(tr, R[p[0]], sizeof(R[0])); memcpy(tr + BLOCK_LEN, R[p[1]], sizeof(R[0]));
        EstimateSpatialCorrMtx(tl, tr, 2, corr, 2 * BLOCK_LEN);
    }
    for (int b = 3; b < 6; b++)
        if (ProcessMVDR(L[b], R[b], BLOCK_LEN, out, 0.0, corr)) fwrite(out, sizeof(short), BLOCK_LEN, w);
    fwrite(corr, sizeof(double), 4, w);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc == 5 && !strcmp(argv[4], "ooo")) {
        FILE *l = fopen(argv[1], "rb"), *r = fopen(argv[2], "rb"), *w = fopen(argv[3], "wb");
        if (!l || !r || !w) return 1;
        const int rc = out_of_order(l, r, w);
        fclose(l); fclose(r); fclose(w);
        return rc;
    }
    if (argc == 5 && !strcmp(argv[4], "vad")) {
        FILE *l = fopen(argv[1], "rb"), *w = fopen(argv[3], "wb");
        if (!l || !w) return 1;
        short blk[BLOCK_LEN];
        while (fread(blk, sizeof(short), BLOCK_LEN, l) == BLOCK_LEN) {
            const unsigned char v = VoiceActivityDetection(blk, BLOCK_LEN) ? 1 : 0;
            fwrite(&v, 1, 1, w);
        }
        fclose(l); fclose(w);
        return 0;
    }
    if (argc != 4) return 1;
    FILE *fpRead1 = fopen(argv[1], "rb"), *fpRead2 = fopen(argv[2], "rb"), *fpWrite = fopen(argv[3], "wb");
    if (!fpRead1 || !fpRead2 || !fpWrite) return 1;
    short rgsInputBufferL[BLOCK_LEN] = {0}, rgsInputBufferR[BLOCK_LEN] = {0}, rgsOutputBuffer[BLOCK_LEN] = {0};
    short rgsTempBufferL[BLOCK_LEN * 2] = {0}, rgsTempBufferR[BLOCK_LEN * 2] = {0};
    double rgdSpatialCorr[2][2] = {{0, 0}, {0, 0}};
    const double dTime = 0;
    int iNumOfIteration = 0;
    while (true) {
        if (fread(rgsInputBufferL, sizeof(short), BLOCK_LEN, fpRead1) == 0) break;
        if (fread(rgsInputBufferR, sizeof(short), BLOCK_LEN, fpRead2) == 0) break;
        if (!VoiceActivityDetection(rgsInputBufferL, BLOCK_LEN)) {
            iNumOfIteration++;
            if (iNumOfIteration > 1) {
                memcpy(rgsTempBufferL + BLOCK_LEN, rgsInputBufferL, sizeof(rgsInputBufferL));
                memcpy(rgsTempBufferR + BLOCK_LEN, rgsInputBufferR, sizeof(rgsInputBufferR));
                EstimateSpatialCorrMtx(rgsTempBufferL, rgsTempBufferR, iNumOfIteration, rgdSpatialCorr, BLOCK_LEN * 2);
            }
            memcpy(rgsTempBufferL, rgsInputBufferL, sizeof(rgsInputBufferL));
            memcpy(rgsTempBufferR, rgsInputBufferR, sizeof(rgsInputBufferR));
        } else {
            iNumOfIteration = 0;
        }
        if (ProcessMVDR(rgsInputBufferL, rgsInputBufferR, BLOCK_LEN, rgsOutputBuffer, dTime, rgdSpatialCorr))
            fwrite(rgsOutputBuffer, sizeof(short), BLOCK_LEN, fpWrite);
    }
    fwrite(rgdSpatialCorr, sizeof(double), 4, fpWrite);
    fclose(fpRead1); fclose(fpRead2); fclose(fpWrite);
    return 0;
}
