// stft_small.hip -- stft.hip compiled a second time for launches WITHOUT a mel plane: 10 frames per workgroup and no mel-table floor on the LDS
// planes (17.7 KB of LDS: 8 workgroups per CU instead of 4).  Only the kernel (stft_small_kernel) and its launcher are built from this unit;
// stft.hip's stft_launch picks it.  Sweep and numbers: the kFR comment in stft.hip, profiles/r03_stft_fr.txt.
#ifndef SE_STFT_SMALL_FR
#define SE_STFT_SMALL_FR 10
#endif
#define SE_STFT_FR SE_STFT_SMALL_FR
#define SE_STFT_MELPLANE 0
#define SE_STFT_TU_SMALL 1
#include "stft.hip"
