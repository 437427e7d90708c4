// raster_host.cpp -- TEST INFRASTRUCTURE: the product's whole raster translation unit (umr_amd/csrc/raster.hip: every kernel, the
// launch sequences and the C-ABI entry points umr_raster_forward / _forward_vis / _backward / umr_raster_workspace_bytes)
// compiled for x86-64 on top of the wave64 emulator of wave_emu.h.  The library this builds takes HOST pointers where
// libumr_hip.so takes device pointers and is loaded by tests/test_raster_library_on_host.py only.
#include "wave_emu.h"
#include "../../umr_amd/csrc/raster.hip"

extern "C" void umr_host_emu_stats(long *blocks, long *switches, long *collectives) {
    *blocks = emu::g_stats.blocks; *switches = emu::g_stats.switches; *collectives = emu::g_stats.collectives;
}
