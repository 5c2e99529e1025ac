// scs_seams.h -- test and tuning seams: environment knobs that change what the library does (small batches, forced kernel
// variants, injected failures, ...).  The PRODUCT library (libscssim_hip.so, bin/scssim) is linked with scs_seams_off.cpp: every
// knob reads as unset and no environment variable of this list is ever looked at.  The tests load libscssim_hip_seams.so
// (SCSSIM_HIP_LIB; bin/scssim_seams) -- the same objects linked with scs_seams_on.cpp, where seam_env is getenv.
//   SCS_TEST_BATCH_SHIFT  pairs per batch = 2^n (many small batches)          SCS_TEST_QK       quality alias rows of 64 / 128 columns
//   SCS_TEST_REDO / _GENERAL / _NO_D1 / SCS_EV_REPLAY  force the read classes' fallbacks     SCS_TEST_SHRINK_OUT  mis-state a batch's text size
//   SCS_READS_SERIAL / SCS_READS_SPLIT / SCS_ERRS_INLINE  launch orders of earlier rounds     SCS_ATTACH_G / SCS_ATTACH_GROUPS  the semi pass as lane groups per template (and how wide)
//   SCS_VMM_FROM_MB / SCS_NO_VMM  where device buffers switch to mapped ranges                SCS_HOST_FASTA / SCS_STAGE_WHOLE  staging paths
//   SCS_TEST_FAIL_AT / SCS_TEST_FAIL_RANK  a CLI rank that dies at a given place
#pragma once
namespace scs { const char* seam_env(const char* name); }
