// storage for the emulator's thread-local "hardware registers" (see hip_emu.h)
#include "hip_emu.h"
thread_local uint3_emu threadIdx, blockIdx;
thread_local dim3 blockDim, gridDim;
thread_local EmuBlock* emu_blk = nullptr;
