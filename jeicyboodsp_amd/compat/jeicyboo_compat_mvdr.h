// jeicyboo_compat_mvdr.h -- BeamForming_MVDR_ver1.cpp's per-block functions (:43-45) over the C ABI.
// A separate library (libjeicyboo_compat_mvdr.so): this program's VoiceActivityDetection has the same name and
// signature as the SS / Wiener one in jeicyboo_compat.h but tests the energy only (:233).
//
// The reference main() (:83-109) drives the three functions in a fixed protocol: VAD on the left block, run
// length, EstimateSpatialCorrMtx on [previous block, block] from the second block of a noise run on, then
// ProcessMVDR with the matrix.  The engine's stream handle (jdsp_mvdr) keeps that whole protocol as its state, so
// here ProcessMVDR feeds the block to the handle -- which runs the same VAD, run length and accumulation -- and
// then writes the handle's matrix into the caller's rgdSpatialCorr; EstimateSpatialCorrMtx itself only checks its
// arguments.  A caller that follows main()'s protocol (the only one the reference has) sees the reference's outputs
// and, after every ProcessMVDR, the reference's matrix.
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
