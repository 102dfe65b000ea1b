// TEST-ONLY: storage for the emulator's thread-locals (see hip_emu.h).
#define LV_EMU_IMPL
#include "hip_emu.h"
