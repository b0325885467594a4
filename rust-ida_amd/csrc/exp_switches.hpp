// Timing-build switches of the kernels, all in one place.
//
// The measurements behind DESIGN.md section 4 were taken with builds of libidahip.so in which one part of a kernel is removed
// or replaced (results are then garbage: such a library only serves tools/kt_libs.sh, tools/newton_time.py, tools/stamps_*.py).
// None of that may reach the shipped library by accident: a switch is only accepted together with -DIDAHIP_TIMING_BUILD, the
// default `make` defines neither (tests/test_abi_symbols.py scans the Makefile and asks the library: idahip_timing_build()
// must return 0), and the kernels only ever read the constants of namespace idahip::tb below -- in a product build every one
// of them is a compile-time `false` / product value and the guarded code folds away.
#pragma once

#if !defined(IDAHIP_TIMING_BUILD) &&                                                                                   \
    (defined(IDAHIP_EXP_NOPRO) || defined(IDAHIP_EXP_NOUPD) || defined(IDAHIP_EXP_NOGATHER) || defined(IDAHIP_EXP_NODIAG) || \
     defined(IDAHIP_EXP_NOSWEEP) || defined(IDAHIP_STAMPS) || defined(IDAHIP_TRAIL_PIPE) || defined(IDAHIP_TRAIL_QUAD) ||   \
     defined(IDAHIP_WP_RING) || defined(IDAHIP_US_PAD) || defined(IDAHIP_SYS_UNR))
#error "IDAHIP_EXP_* / IDAHIP_STAMPS / IDAHIP_TRAIL_* / IDAHIP_WP_RING / IDAHIP_US_PAD / IDAHIP_SYS_UNR are timing-build switches: add -DIDAHIP_TIMING_BUILD (the library then reports itself as one and is not a product)"
#endif

namespace idahip {
namespace tb {

#ifdef IDAHIP_TIMING_BUILD
constexpr bool TIMING_BUILD = true;
#else
constexpr bool TIMING_BUILD = false;
#endif

// lu_trail64w_kernel: no gather and no U12 solve (U12 = constants) / return before the strips / gathered entries = constants
#ifdef IDAHIP_EXP_NOPRO
constexpr bool NOPRO = true;
#else
constexpr bool NOPRO = false;
#endif
#ifdef IDAHIP_EXP_NOUPD
constexpr bool NOUPD = true;
#else
constexpr bool NOUPD = false;
#endif
#ifdef IDAHIP_EXP_NOGATHER
constexpr bool NOGATHER = true;
#else
constexpr bool NOGATHER = false;
#endif
// wg_getrs: the solve without its diagonal blocks / without its sweeps
#ifdef IDAHIP_EXP_NODIAG
constexpr bool NODIAG = true;
#else
constexpr bool NODIAG = false;
#endif
#ifdef IDAHIP_EXP_NOSWEEP
constexpr bool NOSWEEP = true;
#else
constexpr bool NOSWEEP = false;
#endif
// in-kernel time stamps (lu_panelr_kernel's first launch; lu_trail64w_kernel's launch for super-panel k0 == STAMPS): -1 = none
#ifdef IDAHIP_STAMPS
constexpr int STAMPS = IDAHIP_STAMPS;
#else
constexpr int STAMPS = -1;
#endif
// lu_trail64w_kernel: operand reads of step k+1 requested before the arithmetic of step k (product: yes); U12 solve in
// registers on all four waves (product: yes); padding of a row of Us in doubles (product: 2)
#ifdef IDAHIP_TRAIL_PIPE
constexpr bool TRAIL_PIPE = IDAHIP_TRAIL_PIPE != 0;
#else
constexpr bool TRAIL_PIPE = true;
#endif
#ifdef IDAHIP_TRAIL_QUAD
constexpr bool TRAIL_QUAD = IDAHIP_TRAIL_QUAD != 0;
#else
constexpr bool TRAIL_QUAD = true;
#endif
#ifdef IDAHIP_US_PAD
constexpr int US_PAD = IDAHIP_US_PAD;
#else
constexpr int US_PAD = 2;
#endif
// lu_wavepanel_kernel: deep multiplier prefetch rings (product: depth 2 for every slot count)
#ifdef IDAHIP_WP_RING
constexpr bool WP_DEEP_RING = IDAHIP_WP_RING != 0;
#else
constexpr bool WP_DEEP_RING = false;
#endif
// linear_sys_kernel: columns of A and B in flight per thread (product: 16)
#ifdef IDAHIP_SYS_UNR
constexpr int SYS_UNR = IDAHIP_SYS_UNR;
#else
constexpr int SYS_UNR = 16;
#endif

}  // namespace tb
}  // namespace idahip
