// jeicyboo_compat_mvdr.h -- BeamForming_MVDR_ver1.cpp's per-block functions (:43-45) over the C ABI.
// A separate library (libjeicyboo_compat_mvdr.so): this program's VoiceActivityDetection has the same name and
// signature as the SS / Wiener one in jeicyboo_compat.h but tests the energy only (:233).
//
// Each function does what the reference's does, on its own: EstimateSpatialCorrMtx transforms the caller's two
// 1024-sample frames on the GPU and ADDS their contribution to the caller's rgdSpatialCorr (:263-268);
// ProcessMVDR computes its weights from the caller's rgdSpatialCorr (:154-171) and keeps only the reference's own
// statics (the two 511-sample keep buffers and the call counter, :130-131,:137) in a handle.  They can be called
// in main()'s order (:83-109) or in any other.
#ifndef JEICYBOO_COMPAT_MVDR_H
#define JEICYBOO_COMPAT_MVDR_H

#include "../../include/jdsp.h"

bool ProcessMVDR(short *rgsInputBufferL, short *rgsInputBufferR, int iBlockLen, short *rgsOutputBuffer, double dTime,
                 double (*rgdSpatialCorr)[2]);                                                        // :124
bool VoiceActivityDetection(short *rgsInputBuffer, int iFrameCount);                                  // :207 (energy only)
void EstimateSpatialCorrMtx(short *rgsTempBufferL, short *rgsTempBufferR, int iNumOfIteration,
                            double (*rgdSpatialCorr)[2], int iFrameCount);                            // :244
void JeicybooMvdrReset(void);                             // forget the stream (= restarting the program)

#endif
