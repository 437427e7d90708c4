// host_tu.cpp -- TEST INFRASTRUCTURE: one translation unit of the product library (-DUMR_TU='"../../umr_amd/csrc/<name>.hip"')
// compiled for x86-64 on the wave64 emulator of wave_emu.h.  tests/host_raster.py builds all of them into libumr_host.so,
// the whole C ABI of include/umr_hip.h on host pointers, for the CPU tests.
#include "wave_emu.h"
#include UMR_TU
#ifdef UMR_TU_STATS
extern "C" void umr_host_emu_stats(long *blocks, long *switches, long *collectives) {
    *blocks = emu::g_stats.blocks; *switches = emu::g_stats.switches; *collectives = emu::g_stats.collectives;
}
#endif
