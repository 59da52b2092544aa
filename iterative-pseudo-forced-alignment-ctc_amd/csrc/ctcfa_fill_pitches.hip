// The fill kernel's instantiations for one group of vocabulary pitches (-DCTCFA_PITCH_GROUP=1..7): see
// ctcfa_fill_select.hip.h.  Compiled by __graft_entry__.build(), one hipcc process per group.
#include <hip/hip_runtime.h>
#define CTCFA_FILL_KERNEL_ONLY
#include "ctcfa_fill_select.hip.h"

#ifndef CTCFA_PITCH_GROUP
#define CTCFA_PITCH_GROUP 0   // (a bare `hipcc -c` of this file: an empty group)
#endif

#define CTCFA_GROUP_FN_(n) ctcfa_fill_group_##n
#define CTCFA_GROUP_FN(n) CTCFA_GROUP_FN_(n)

FillFn CTCFA_GROUP_FN(CTCFA_PITCH_GROUP)(int K, int VP, bool ck) {
    switch (VP) {
#if CTCFA_PITCH_GROUP == 1
        case 40: return fill_any<40>(K, ck);
#elif CTCFA_PITCH_GROUP == 2
        case 48: return fill_any<48>(K, ck);
#elif CTCFA_PITCH_GROUP == 3
        case 56: return fill_any<56>(K, ck);
#elif CTCFA_PITCH_GROUP == 4
        case 64: return fill_any<64>(K, ck);
#elif CTCFA_PITCH_GROUP == 5
        case 80: return fill_any<80>(K, ck);
        case 96: return fill_any<96>(K, ck);
        case 112: return fill_any<112>(K, ck);
#elif CTCFA_PITCH_GROUP == 6
        case 128: return fill_any<128>(K, ck);
        case 160: return fill_any<160>(K, ck);
#elif CTCFA_PITCH_GROUP == 7
        case 192: return fill_any<192>(K, ck);
        case 256: return fill_any<256>(K, ck);
#endif
        default: return nullptr;
    }
}
