// compat_mvdr_selftest.cpp -- BeamForming_MVDR_ver1.cpp's main() loop (:83-109) in structure, on the per-block
// functions of jeicyboo_compat_mvdr.h.  usage: compat_mvdr_selftest left.raw right.raw out.bin
// out.bin: the int16 blocks the loop writes, then the final rgdSpatialCorr (4 doubles).
#include <cstdio>
#include <cstring>

#include "jeicyboo_compat_mvdr.h"

#define BLOCK_LEN 512

int main(int argc, char **argv)
{
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
