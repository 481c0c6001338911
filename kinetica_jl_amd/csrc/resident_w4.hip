// The resident integrator built for two workgroups per compute unit (128 VGPRs per lane): see resident.hip.
#define RES_WAVES_PER_EU 4
#include "resident.hip"
