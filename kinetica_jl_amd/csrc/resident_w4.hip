// The resident integrator built for two workgroups per compute unit (128 VGPRs per lane, phases inlined): see resident.hip.
#define RES_WAVES_PER_EU 4
#define RES_PHASE __device__ __forceinline__
#include "resident.hip"
