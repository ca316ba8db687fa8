// The one translation unit that carries the emulator's context-switch routine (test build only).
#define HIPEMU_IMPLEMENTATION
#include "hip_emu.h"
