/* Umbrella header, same path as the reference's lib/include/wfa_gpu.h:25-30. */
#ifndef WFA_GPU_H
#define WFA_GPU_H
#include "../../../include/wfa_gpu_abi.h"
#include "../../../include/wfa_gpu_device.h"
#endif
