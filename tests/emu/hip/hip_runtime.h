// TEST INFRASTRUCTURE: shadows <hip/hip_runtime.h> when the kernels are compiled for the host
// emulation (g++ -I tests/emu); see ../hip_emu.h.
#pragma once
#include "../hip_emu.h"
