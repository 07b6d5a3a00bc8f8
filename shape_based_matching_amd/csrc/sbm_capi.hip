// sbm_capi.hip — host side of libsbm_hip.so: the C ABI of include/sbm.h on top
// of the kernels in sbm_kernels.h.  gfx950 only; there is no CPU fallback: every
// entry point fails with SBM_ERR_HIP when no GPU is usable.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <dlfcn.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../include/sbm.h"
#include "sbm_kernels.h"
#include "sbm_quantize_stream.h"
#include "sbm_train_kernels.h"

using namespace sbm;

// Round 3: the host side is one translation unit in four parts.
#include "sbm_capi_ctx.inc"     // sbm_ctx, buffers, launch helpers
#include "sbm_capi_match.inc"   // the match entry points of include/sbm.h
#include "sbm_capi_stages.inc"  // stage entry points, profiling
#include "sbm_capi_multi.inc"   // RCCL, sharded / banded steps, sbm_match_sharded
