// common.h — host-side constants and types of the bhSPARSE facade.
// Mirrors SpGEMM_cuda/common.h:26-37 (BHSPARSE_SUCCESS, index_type, value_type,
// NUM_PLATFORMS and the platform slots) with a new BHSPARSE_HIP slot; no CUDA /
// OpenCL / HIP headers are needed by host code (the device side is reached only
// through the C-ABI of include/bhsparse_hip.h).
#ifndef BHSPARSE_AMD_COMMON_H
#define BHSPARSE_AMD_COMMON_H

#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>

#define BHSPARSE_SUCCESS 0

typedef int    index_type;
#ifdef BHS_VALUE_FLOAT          // float build: compile with -DBHS_VALUE_FLOAT and link libbhsparse_hip_f32.so
typedef float  value_type;
#else
typedef double value_type;
#endif

#define NUM_PLATFORMS   9
#define NAIVE           0
#define BHSPARSE_CUDA   1   // accepted as an alias of the HIP backend (published command lines keep working)
#define BHSPARSE_OPENCL 2   // alias
#define BHSPARSE_HIP    3   // new slot (3..8 are unused in the reference)

#endif
