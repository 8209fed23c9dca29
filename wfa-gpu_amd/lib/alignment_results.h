/* Compatibility header: the reference's name, this build's single ABI header. */
#include "../../include/wfa_gpu_abi.h"
