// Which instantiation of ctcfa::fill_kernel<K, VP, CK> serves a plan.  Shared by ctcfa.hip and ctcfa_fill_pitches.hip:
// __graft_entry__.build() compiles the vocabulary pitches other than 32 as translation units of their own, in parallel
// (one hipcc over everything takes three minutes), and ctcfa.hip asks the group that holds a pitch.
#pragma once
#include "ctcfa_kernels.hip.h"

using FillFn = void (*)(const ctcfa::SegDesc*, const float*, const int32_t*, uint32_t*, float*, int, int, int,
                        const ctcfa::FillRoles*, const ctcfa::WatchDesc*, int32_t*, int);

template <int VP, bool CK>
FillFn fill_for_k(int K) {
    switch (K) {
        case 1: return ctcfa::fill_kernel<1, VP, CK>;
        case 2: return ctcfa::fill_kernel<2, VP, CK>;
        case 3: return ctcfa::fill_kernel<3, VP, CK>;
        case 4: return ctcfa::fill_kernel<4, VP, CK>;
        case 5: return ctcfa::fill_kernel<5, VP, CK>;
        case 6: return ctcfa::fill_kernel<6, VP, CK>;
        case 8: return ctcfa::fill_kernel<8, VP, CK>;
        case 10: return ctcfa::fill_kernel<10, VP, CK>;
        case 12: return ctcfa::fill_kernel<12, VP, CK>;
        case 16: return ctcfa::fill_kernel<16, VP, CK>;
        default: return nullptr;
    }
}

template <int VP>
FillFn fill_any(int K, bool ck) {
    if constexpr (VP <= 64) {
        if (ck) return fill_for_k<VP, true>(K);
    }
    return fill_for_k<VP, false>(K);
}

// (defined in ctcfa_fill_pitches.hip, one per -DCTCFA_PITCH_GROUP)
FillFn ctcfa_fill_group_1(int K, int VP, bool ck);   // 40
FillFn ctcfa_fill_group_2(int K, int VP, bool ck);   // 48
FillFn ctcfa_fill_group_3(int K, int VP, bool ck);   // 56
FillFn ctcfa_fill_group_4(int K, int VP, bool ck);   // 64
FillFn ctcfa_fill_group_5(int K, int VP, bool ck);   // 80, 96, 112
FillFn ctcfa_fill_group_6(int K, int VP, bool ck);   // 128, 160
FillFn ctcfa_fill_group_7(int K, int VP, bool ck);   // 192, 256
