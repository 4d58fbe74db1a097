// TEST INFRASTRUCTURE ONLY: stands in for <hip/hip_runtime.h> when the kernel
// sources under dsrc_amd/csrc are compiled for the CPU by tests/emu/Makefile.
#pragma once
#include "../../emu_hip.h"
