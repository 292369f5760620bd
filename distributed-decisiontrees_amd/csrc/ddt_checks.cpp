// ddt_checks.cpp -- what the build knows about its own device code, baked into libddt.so (not a marker file next to the library,
// which a copy or an install can lose).  __graft_entry__.build() compiles the library, runs the two instruction-level checks on the result
//   tools/check_s2_isa.py      no instruction touches the "_s2" kernels' node-record SGPRs while their scalar loads are in flight
//   tools/check_dma_waits.py   every counted `s_waitcnt vmcnt(N)` a barrier relies on covers the chunk DMA on every path (the deep kernels)
// writes their outcome into lib/checks.flags and relinks with this file recompiled (the device code does not change).  A build that could
// not run a check (no disassembler) keeps the 0: the automatic kernel choice then avoids the kernels concerned (csrc/ddt_choice.cpp
// s2_disabled / deep_disabled) and ddt_info::build_checks says so.
#ifndef DDT_S2_CHECKED
#define DDT_S2_CHECKED 0
#endif
#ifndef DDT_DMA_CHECKED
#define DDT_DMA_CHECKED 0
#endif
extern "C" {
extern const int ddt_build_s2_checked = DDT_S2_CHECKED;
extern const int ddt_build_dma_checked = DDT_DMA_CHECKED;
}
