// Build-time knobs of libdd3d_hip.so: every -DDD3D_...=... a translation unit was compiled with, as a string.
//
// The product build (__graft_entry__.build()) passes none: dd3d_build_flags() then returns "" and dd3d_amd/hip.py loads the library.  A
// library built with any of them -- the A/B variants of tests/tools/build_variant.sh; all of them compute CORRECT results, the knobs trade
// ring depths / schedules / store forms -- says so, and hip.py refuses it unless the caller selected it explicitly (DD3D_HIP_LIB /
// DD3D_ALLOW_VARIANT_LIB=1), so that a stray -D cannot ship as the product library unnoticed (round-4 verdict).  Timing experiments that
// compute WRONG results (ablations of the K loop, racy barriers) are not in these sources at all: tests/tools/variants/*.patch.
//
// Included by common.h BEFORE any translation unit gives a knob its default (`#ifndef X / #define X default`), so "defined here" means
// "given on the command line".
#pragma once
#define DD3D_BF_STR2(x) #x
#define DD3D_BF_STR(x) DD3D_BF_STR2(x)
#ifdef DD3D_EPI_LDS
#define DD3D_BF_0 " DD3D_EPI_LDS=" DD3D_BF_STR(DD3D_EPI_LDS)
#else
#define DD3D_BF_0 ""
#endif
#ifdef DD3D_EPI_T
#define DD3D_BF_1 " DD3D_EPI_T=" DD3D_BF_STR(DD3D_EPI_T)
#else
#define DD3D_BF_1 ""
#endif
#ifdef DD3D_LDS_KIB_4W
#define DD3D_BF_2 " DD3D_LDS_KIB_4W=" DD3D_BF_STR(DD3D_LDS_KIB_4W)
#else
#define DD3D_BF_2 ""
#endif
#ifdef DD3D_LDS_KIB_8W
#define DD3D_BF_3 " DD3D_LDS_KIB_8W=" DD3D_BF_STR(DD3D_LDS_KIB_8W)
#else
#define DD3D_BF_3 ""
#endif
#ifdef DD3D_PREFETCH_DISTANCE
#define DD3D_BF_4 " DD3D_PREFETCH_DISTANCE=" DD3D_BF_STR(DD3D_PREFETCH_DISTANCE)
#else
#define DD3D_BF_4 ""
#endif
#ifdef DD3D_PRODUCER_WAVES
#define DD3D_BF_5 " DD3D_PRODUCER_WAVES=" DD3D_BF_STR(DD3D_PRODUCER_WAVES)
#else
#define DD3D_BF_5 ""
#endif
#ifdef DD3D_ROW_LDS_KIB_4W
#define DD3D_BF_6 " DD3D_ROW_LDS_KIB_4W=" DD3D_BF_STR(DD3D_ROW_LDS_KIB_4W)
#else
#define DD3D_BF_6 ""
#endif
#ifdef DD3D_ROW_LDS_KIB_8W
#define DD3D_BF_7 " DD3D_ROW_LDS_KIB_8W=" DD3D_BF_STR(DD3D_ROW_LDS_KIB_8W)
#else
#define DD3D_BF_7 ""
#endif
#ifdef DD3D_ROW_NSA_MAX
#define DD3D_BF_8 " DD3D_ROW_NSA_MAX=" DD3D_BF_STR(DD3D_ROW_NSA_MAX)
#else
#define DD3D_BF_8 ""
#endif
#ifdef DD3D_ROW_NSB_MAX
#define DD3D_BF_9 " DD3D_ROW_NSB_MAX=" DD3D_BF_STR(DD3D_ROW_NSB_MAX)
#else
#define DD3D_BF_9 ""
#endif
#ifdef DD3D_SCHED_VARIANT
#define DD3D_BF_10 " DD3D_SCHED_VARIANT=" DD3D_BF_STR(DD3D_SCHED_VARIANT)
#else
#define DD3D_BF_10 ""
#endif
#ifdef DD3D_STEM_TW
#define DD3D_BF_11 " DD3D_STEM_TW=" DD3D_BF_STR(DD3D_STEM_TW)
#else
#define DD3D_BF_11 ""
#endif
#ifdef DD3D_STEM_WAVES
#define DD3D_BF_12 " DD3D_STEM_WAVES=" DD3D_BF_STR(DD3D_STEM_WAVES)
#else
#define DD3D_BF_12 ""
#endif
#ifdef DD3D_STEM_CHAINS
#define DD3D_BF_13 " DD3D_STEM_CHAINS=" DD3D_BF_STR(DD3D_STEM_CHAINS)
#else
#define DD3D_BF_13 ""
#endif
#ifdef DD3D_CHAIN_A_AUX
#define DD3D_BF_14 " DD3D_CHAIN_A_AUX=" DD3D_BF_STR(DD3D_CHAIN_A_AUX)
#else
#define DD3D_BF_14 ""
#endif
#ifdef DD3D_CHAIN_ACQUIRE
#define DD3D_BF_15 " DD3D_CHAIN_ACQUIRE=" DD3D_BF_STR(DD3D_CHAIN_ACQUIRE)
#else
#define DD3D_BF_15 ""
#endif
#ifdef DD3D_CHAIN_RES_SC1
#define DD3D_BF_16 " DD3D_CHAIN_RES_SC1=" DD3D_BF_STR(DD3D_CHAIN_RES_SC1)
#else
#define DD3D_BF_16 ""
#endif
#ifdef DD3D_CHAIN_B_FIRST
#define DD3D_BF_17 " DD3D_CHAIN_B_FIRST=" DD3D_BF_STR(DD3D_CHAIN_B_FIRST)
#else
#define DD3D_BF_17 ""
#endif
#ifdef DD3D_ROW_STAMP
#define DD3D_BF_18 " DD3D_ROW_STAMP=" DD3D_BF_STR(DD3D_ROW_STAMP)
#else
#define DD3D_BF_18 ""
#endif
#ifdef DD3D_ROW_B_WAVES
#define DD3D_BF_19 " DD3D_ROW_B_WAVES=" DD3D_BF_STR(DD3D_ROW_B_WAVES)
#else
#define DD3D_BF_19 ""
#endif
#ifdef DD3D_ROW_B_SADDR
#define DD3D_BF_20 " DD3D_ROW_B_SADDR=" DD3D_BF_STR(DD3D_ROW_B_SADDR)
#else
#define DD3D_BF_20 ""
#endif
#ifdef DD3D_ROW_B_WAVES_HI
#define DD3D_BF_21 " DD3D_ROW_B_WAVES_HI=" DD3D_BF_STR(DD3D_ROW_B_WAVES_HI)
#else
#define DD3D_BF_21 ""
#endif
#ifdef DD3D_ROW_PRIO_SLICE
#define DD3D_BF_22 " DD3D_ROW_PRIO_SLICE=" DD3D_BF_STR(DD3D_ROW_PRIO_SLICE)
#else
#define DD3D_BF_22 ""
#endif
#define DD3D_BUILD_FLAGS DD3D_BF_0 DD3D_BF_1 DD3D_BF_2 DD3D_BF_3 DD3D_BF_4 DD3D_BF_5 DD3D_BF_6 DD3D_BF_7 DD3D_BF_8 DD3D_BF_9 DD3D_BF_10 DD3D_BF_11 DD3D_BF_12 DD3D_BF_13 DD3D_BF_14 DD3D_BF_15 DD3D_BF_16 DD3D_BF_17 DD3D_BF_18 DD3D_BF_19 DD3D_BF_20 DD3D_BF_21 DD3D_BF_22
