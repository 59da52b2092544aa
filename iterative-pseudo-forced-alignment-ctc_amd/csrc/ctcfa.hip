// ctcfa.hip -- C ABI (include/ctcfa.h) of the gfx950 CTC forced-alignment engine.
//
// Host side of the hot path: batch geometry ("plan"), workspace in HBM, kernel
// selection and launch.  No CPU compute path exists in this library: every entry
// that aligns needs a HIP device.  Reference interfaces replaced: see include/ctcfa.h.
#include "ctcfa_kernels.hip.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <hip/hip_ext.h>

#include "../../include/ctcfa.h"
#include "ctcfa_fill_select.hip.h"

using ctcfa::BtParams;
using ctcfa::SegDesc;

// Grow-only device scratch of the host-buffer entry (ctcfa_align_batch is synchronous, so one
// call's buffers are free again when the next one starts): an anchor iteration issues hundreds
// of small calls and would otherwise pay ~15 hipMalloc/hipFree pairs in each.
struct ScratchSlot {
    void* p = nullptr;
    size_t cap = 0;
};
enum { kSlotLpz, kSlotIn, kSlotOut, kSlotSegs, kSlotRoles, kSlotBits, kSlotLastcol, kSlotCompact, kSlotWinTable, kSlotWinOffs, kSlotWinList, kNumSlots };

struct ctcfa_engine {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;
    int lds_limit = 160 * 1024;
    int num_cu = 256;
    ScratchSlot scratch[kNumSlots];
    // pinned staging of the host-buffer entries: ONE packed upload (role table, segment table, labels,
    // utterance starts) and ONE result download per call
    unsigned char* h_in = nullptr;
    size_t h_in_cap = 0;
    unsigned char* h_out = nullptr;
    size_t h_out_cap = 0;
    int64_t trace_ns[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // CTCFA_CALL_TRACE
    int64_t trace_calls = 0;
    int32_t run_seq = 0;   // runs are numbered (1, 2, ...): what a fill writes into its workspace's error word when a wait gives up
    // completion word of small host-buffer calls (BtArgs::done_word): a device counter that only grows, a pinned word the last
    // workgroup of a call's backtrack writes the call's number to, and what the host expects of both
    uint32_t* d_done_count = nullptr;
    uint32_t* h_done = nullptr;        // pinned, host-coherent
    uint32_t* d_done_word = nullptr;   // ... as the device sees it
    uint32_t done_total = 0, done_seq = 0;
};

// Workspaces (trace words + last-column scores) a plan rotates through in the pipelined entry.  Two would do for
// "backtrack(k) beside fill(k+1)"; with four, the backtrack that last read the workspace of run k is that of run
// k - 4, long finished -- the HOST can see that (or wait for it) and no wait has to sit in the GPU queue between
// two fills (such a packet, satisfied or not, cost config 3 12 us per step).
constexpr int kWorkspaces = 4;
constexpr int kMaxScoreLength = 1 << 20;   // score_min_mean_over_L (frames): anything a window can hold

struct ctcfa_plan {
    ctcfa_engine* eng = nullptr;
    ctcfa_params prm{};
    int B = 0, V = 0, K = 0, W = 0, VP = 0;
    int lds_fill = 0, lds_bt = 0, rec_bytes = 0, lab_bytes = 0, nblk_max = 0;
    bool overlap_ok = true;   // a backtrack workgroup of the previous run fits a CU beside this plan's fill workgroups
    bool ckpt = false;  // fill stores table rows, the backtrack recomputes its decisions (V <= 64)
    int bt_waves = 4;   // waves of a backtrack workgroup (checkpoint mode: striders, one 32-row block each)
    int bt_scorers = 0; // ... plus waves that only score utterances (checkpoint mode, plans with utterances)
    int fol_bytes = 0;  // checkpoint mode: LDS copy of frame_of_label
    bool have_utt = false;
    std::vector<SegDesc> segs;
    int64_t total_T = 0, total_C = 0, total_U = 0, bits_words = 0, alg_bytes = 0;
    SegDesc* d_segs = nullptr;
    bool scratch_owned = false;  // d_segs / d_roles / d_bits[0] / d_lastcol[0] live in the engine's scratch
    int VPbt = 0;                       // pitch of the checkpoint-mode backtrack (the vocabulary's own; VP is the FILL's: 32 for a narrowed plan)
    uint32_t done_target = 0, done_value = 0;   // this run's backtrack writes the engine's completion word (0 / 0: no)
    bool narrow = false;                // narrowed plan (ctcfa::narrow_build): fill and backtrack stage the 32 entries each segment's text uses
    // workspaces: index 0 always; the others exist once the pipelined entry has been used
    uint32_t* d_bits[kWorkspaces] = {nullptr, nullptr, nullptr, nullptr};
    float* d_lastcol[kWorkspaces] = {nullptr, nullptr, nullptr, nullptr};
    // pipelined mode: backtrack of run k on `side` overlaps the fill of run k+1 on the caller's stream
    hipStream_t side = nullptr;
    hipEvent_t ev_fill_done[kWorkspaces] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_bt_done[kWorkspaces] = {nullptr, nullptr, nullptr, nullptr};
    bool bt_pending[kWorkspaces] = {false, false, false, false};
    hipEvent_t bt_done_ev[kWorkspaces] = {nullptr, nullptr, nullptr, nullptr};  // the event that marks workspace q free again
    int64_t pipe_runs = 0;
    // event ring: 4 events per recorded run (fill start/end, backtrack start/end)
    std::vector<hipEvent_t> ev;
    int ev_slots = 0;
    int64_t ev_runs = 0;
    int ev_stride = 1;       // record timing events on every ev_stride-th run
    int64_t run_counter = 0;
    void (*fill_fn)(const SegDesc*, const float*, const int32_t*, uint32_t*, float*, int, int, int,
                    const ctcfa::FillRoles*, const ctcfa::WatchDesc*, int32_t*, int) = nullptr;
    int32_t last_run[kWorkspaces] = {0, 0, 0, 0};   // number of the run that last filled workspace q (its backtrack looks for it in the error word)
    // shared fills: watch columns of every group, the widest group, unique emission frames
    std::vector<ctcfa::WatchDesc> watch;
    ctcfa::WatchDesc* d_watch = nullptr;
    int nwatch_max = 0;
    int S = 1;                        // label width (> 1: multi-character tokens, every segment through windowed_kernel)
    int n_fill = 0;                   // segments that get a fill workgroup of their own
    int n_emission_blocks = 0;
    int64_t total_lpz_T = 0;
    // windowed regime (T > min_window_size): segment indices, fp32 table + per-column offsets
    std::vector<int32_t> win_list;
    int32_t* d_win_list = nullptr;
    float* d_win_table = nullptr;
    int32_t* d_win_offs = nullptr;
    int64_t win_table_floats = 0, win_cols = 0;
    int lds_win = 0;
    int win_cmax = 0;   // most label columns of a windowed segment (band_fill_kernel's columns per lane)
    bool gather = false;              // wide vocabulary: fill_gather_kernel
    ctcfa::FillRoles roles{};         // what each wave of a fill workgroup does
    ctcfa::FillRoles* d_roles = nullptr;
};

namespace {

thread_local std::string g_err;

int set_err(ctcfa_engine* e, int code, const std::string& msg) {
    if (e) e->err = msg;
    g_err = msg;
    return code;
}

// The engine's device for the duration of a call; the caller's (torch's) current device comes back after it.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    bool ok = true;   // false: the engine's device could not be made current (the entry point reports CTCFA_ERR_HIP)
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) ok = false;
        else if (prev != device) ok = switched = (hipSetDevice(device) == hipSuccess);
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define GUARD_DEVICE(eng)                                                                         \
    DeviceGuard on_device((eng)->device);                                                        \
    if (!on_device.ok) return set_err(eng, CTCFA_ERR_HIP, "hipSetDevice: the engine's device could not be made current")

#define HIP_TRY(eng, expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess)                                                                \
            return set_err(eng, CTCFA_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

FillFn select_fill(int K, int VP, bool ck) {
    switch (VP) {
        case 32: return fill_any<32>(K, ck);
#if defined(CTCFA_DEV_VP32_ONLY)   // tuning builds (tools/build_variant.sh): one pitch compiles in a fifth of the time
#if defined(CTCFA_DEV_WIDE)        // (... of the wide pitches)
        case 128: return fill_any<128>(K, ck);
        case 256: return fill_any<256>(K, ck);
#endif
#elif defined(CTCFA_SPLIT_BUILD)   // __graft_entry__.build(): the other pitches are compiled beside this file, in parallel
        case 40: return ctcfa_fill_group_1(K, VP, ck);
        case 48: return ctcfa_fill_group_2(K, VP, ck);
        case 56: return ctcfa_fill_group_3(K, VP, ck);
        case 64: return ctcfa_fill_group_4(K, VP, ck);
        case 80: case 96: case 112: return ctcfa_fill_group_5(K, VP, ck);
        case 128: case 160: return ctcfa_fill_group_6(K, VP, ck);
        case 192: case 256: return ctcfa_fill_group_7(K, VP, ck);
#else                              // one translation unit (hipcc ctcfa.hip): everything here
        case 40: return fill_any<40>(K, ck);
        case 48: return fill_any<48>(K, ck);
        case 56: return fill_any<56>(K, ck);
        case 64: return fill_any<64>(K, ck);
        case 80: return fill_any<80>(K, ck);
        case 96: return fill_any<96>(K, ck);
        case 112: return fill_any<112>(K, ck);
        case 128: return fill_any<128>(K, ck);
        case 160: return fill_any<160>(K, ck);
        case 192: return fill_any<192>(K, ck);
        case 256: return fill_any<256>(K, ck);
#endif
        default: return nullptr;
    }
}

using StrideFn = void (*)(ctcfa::BtArgs);
StrideFn select_strider(int VP, bool narrow = false) {
    if (narrow) {
        switch (VP) {
            case 32: return ctcfa::stride_backtrack_kernel<32, true>;
#ifndef CTCFA_DEV_VP32_ONLY
            case 64: return ctcfa::stride_backtrack_kernel<64, true>;
#endif
            default: return nullptr;
        }
    }
    switch (VP) {
        case 32: return ctcfa::stride_backtrack_kernel<32>;
#ifndef CTCFA_DEV_VP32_ONLY
        case 40: return ctcfa::stride_backtrack_kernel<40>;
        case 48: return ctcfa::stride_backtrack_kernel<48>;
        case 56: return ctcfa::stride_backtrack_kernel<56>;
        case 64: return ctcfa::stride_backtrack_kernel<64>;
#endif
        default: return nullptr;
    }
}

// checkpoint-mode backtrack: waves per workgroup (each recomputes and walks one 32-row block at a time)
// A launch that leaves CUs idle anyway (at most one segment per CU: a window of the anchor iteration, a round of a few
// files in lockstep) gives every workgroup seven striders -- more hedged windows per turn, nothing beside them to slow
// down (T = 499: backtrack 15.1 -> 12.8 us, T = 700: 14.6 -> 11.8; profiles/r03_small_window_modes.txt).
int strider_waves(bool lone) {
    int n = lone ? 7 : 3;   // (beside the fill of the next batch more striders cost the fill more than they save: 3 + 1 scorer = 4 waves, one per SIMD)
    if (const char* e = std::getenv("CTCFA_SB_WAVES")) n = std::atoi(e);
    return std::max(1, std::min(ctcfa::kSbMaxWaves - 1, n));   // (one more wave may join them as a scorer)
}

// dynamic LDS of one checkpoint-mode backtrack workgroup: rec | labels | frame_of_label | (a narrowed plan: ring entry ->
// vocabulary entry, 128 bytes) | -inf column + NW slots | char_probs
int lds_bytes_strider(int rec_bytes, int lab_bytes, int VP, int nw, int T, bool narrow) {
    return rec_bytes + lab_bytes + 4 * lab_bytes + (narrow ? VP * 4 : 0) + (nw + 1) * ctcfa::kRows * VP * 4 + T * 4;
}

// columns a tile of K columns per lane adds to the trellis (its halo lanes are copies)
int tile_useful_cols(int K) { return (64 - ctcfa::halo_lanes(K)) * K; }

// Role table.  Workgroups of 4 or 8 waves land evenly on the four SIMDs of a CU ({w, w+4} share
// one, and the neighbour workgroup on the same CU is rotated by one SIMD: tools/hwid_probe.hip), so
// small tile counts are padded with waves that leave at once: tiles first one per SIMD (waves
// 0..3), then the second wave of SIMD pairs 0 and 1 (waves 4, 6); the producer(s) take the second
// wave of the other pairs (waves 5, 7) -- with the rotation every SIMD of the CU ends up with the
// same number of tiles for W = 4 and W = 6.
ctcfa::FillRoles tile_roles(int K, int W, int NS, int nprod) {
    ctcfa::FillRoles r{};
    const int U = tile_useful_cols(K);
    const int XW = ctcfa::halo_lanes(K) * K;
    r.nstages = W;
    r.cpad = W * U;
    r.nslots = NS;
    r.nprod = nprod;
    int order[16];
    bool padded8 = false;
    if (W + nprod <= 4) {
        r.nwaves = 4;
        for (int i = 0; i < 4; ++i) order[i] = i;
    } else if (W + nprod <= 8 && W <= 6) {
        r.nwaves = 8;
        padded8 = true;
        const int o[8] = {0, 1, 2, 3, 4, 6, 5, 7};
        for (int i = 0; i < 8; ++i) order[i] = o[i];
    } else {
        r.nwaves = W + nprod;
        for (int i = 0; i < r.nwaves; ++i) order[i] = i;
    }
    // Issue priorities (s_setprio): the later half of a segment's tiles one above the earlier half, the producers below
    // both and above the previous batch's striders.  CTCFA_TILE_PRIOS="p0,p1,.." / CTCFA_PROD_PRIO=p
    // (tuning): per-tile priorities, the last one repeated.
    int tile_prio[16], prod_prio = CTCFA_PRODUCER_PRIO;
    for (int w = 0; w < 16; ++w) tile_prio[w] = CTCFA_TILE_PRIO_BASE + (w >= (W + 1) / 2 ? 1 : 0);
    if (const char* e = std::getenv("CTCFA_TILE_PRIOS")) {
        int last = tile_prio[0], w = 0;
        for (const char* c = e; w < 16; ++w) {
            if (*c) { last = std::atoi(c); while (*c && *c != ',') ++c; if (*c == ',') ++c; }
            tile_prio[w] = last;
        }
    }
    if (const char* e = std::getenv("CTCFA_PROD_PRIO")) prod_prio = std::atoi(e);
    auto prio = [](int p) { return std::max(0, std::min(3, p)); };
    for (int i = 0; i < r.nwaves; ++i) r.wave[i] = {ctcfa::kRoleIdle, 0, 0};
    for (int w = 0; w < W; ++w)
        r.wave[order[w]] = {ctcfa::role_with_prio(ctcfa::kRoleTile, prio(tile_prio[w])), (int8_t)w, (int16_t)(w * U - XW)};
    const int8_t prole = ctcfa::role_with_prio(ctcfa::kRoleProducer, prio(prod_prio));
    if (padded8) {   // producers: second wave of SIMD pairs 2 and 3
        r.wave[5] = {prole, 0, 0};
        if (nprod == 2) r.wave[7] = {prole, 1, 0};
    } else {
        for (int p = 0; p < nprod; ++p) r.wave[order[W + p]] = {prole, (int8_t)p, 0};
    }
    return r;
}

const int kKs[] = {1, 2, 3, 4, 5, 6, 8, 10, 12, 16};

int waves_of(int W, int nprod) {
    if (W + nprod <= 4) return 4;
    if (W + nprod <= 8 && W <= 6) return 8;
    return W + nprod;
}

// launch bounds of fill_kernel<K, ..>: 16 waves; K >= 10: 320 threads, K == 8: 512
bool shape_launchable(int K, int W, int nprod) {
    const int waves_per_wg = waves_of(W, nprod);
    return !(W > 16 - nprod || (K >= 10 && waves_per_wg > 5) || (K == 8 && waves_per_wg > 8));
}

int lds_bytes_fill(int NS, int W, int K, int VP, int nwatch = 0) {
    // emission ring (NS slots) + exchange rings + last-column ring + counters + sink
    // (+ shared fills: a ring and a target offset per watch column)
    // (above 80 entries the ring holds e alone: VP + 4 floats a row)
    const int row_bytes = VP > 80 ? (VP + 4) * 4 : (VP + ctcfa::kPitchPad) * 8;
    return NS * ctcfa::kRows * row_bytes + W * ctcfa::kExchangeRing * ctcfa::halo_lanes(K) * K * 4 +
           64 * 4 + ctcfa::kFlagInts * 4 + ctcfa::kSinkBytes + nwatch * (64 * 4 + 8);
}

int roundup(int x, int m) { return (x + m - 1) / m * m; }

int vocab_pitch(int vocab) {
    return vocab <= 32 ? 32 : vocab <= 40 ? 40 : vocab <= 48 ? 48 : vocab <= 56 ? 56 : vocab <= 64 ? 64
           : vocab <= 80 ? 80 : vocab <= 96 ? 96 : vocab <= 112 ? 112 : vocab <= 128 ? 128 : vocab <= 160 ? 160 : vocab <= 192 ? 192 : 256;
}
constexpr int kMaxStagedVocab = 256;   // wider vocabularies: the compact matrix of a call's own labels, or the gather kernel

// Label columns the widest launch shape of the fill kernel covers (more: status CTCFA_ST_TEXT_TOO_LONG for that
// segment).  The same rules as pick_shape: tile widths and wave counts of the launch bounds, a ring of 4 slots
// or, where only that fits the LDS, 3.
int label_column_limit(int lds_limit, int VP, int nprod, bool gather) {
    if (gather) return 15 * 64 + 1;
    int c_limit = 0;
    for (int K : kKs) {
        for (int W = 16 - nprod; W >= 1; --W) {
            if (!shape_launchable(K, W, nprod) || lds_bytes_fill(3, W, K, VP) > lds_limit) continue;
            c_limit = std::max(c_limit, W * tile_useful_cols(K) - (K - 1));
            break;
        }
    }
    return c_limit;
}

// Launch-shape choice from a two-bound cost model of the fill kernel (constants fitted
// to tools/shape_search2.py sweeps, cycles per trellis row):
//   one tile (wave) alone needs                62 + 21.4 K   (its own serial instruction issue)
//   a SIMD retires a row of a K-column tile in  5 + 15 K     x tiles resident on the SIMD
//   (decision-word mode: 6 more vector instructions per cell on both bounds);
//   G workgroups share a CU (LDS, wave slots, VGPRs); a batch takes ceil(B / (G x CUs)) rounds;
//   wide tiles lose a little more to contention than the bounds say (factor 1 + 0.03 K), and shapes
//   that leave no room for two backtrack workgroups of the previous batch beside them pay 10 %.
struct ShapeChoice { int K, W, NS; };

int vgprs_of(int K, bool ckpt) {  // compiled register counts, rounded up to the allocation granule 8
    switch (K) {
        case 1: case 2: return 64;
        case 3: return ckpt ? 64 : 80;
        case 4: return ckpt ? 88 : 96;
        case 5: return ckpt ? 104 : 112;
        case 6: return ckpt ? 112 : 128;
        case 8: return ckpt ? 144 : 160;
        case 10: return ckpt ? 168 : 192;
        case 12: return ckpt ? 200 : 224;
        default: return 256;
    }
}
constexpr int kBacktrackVgprs = 96;   // backtrack_kernel<*>: __launch_bounds__(256, 5)

bool pick_shape(int B, int Cmax, int VP, int lds_limit, int lds_beside, int num_cu, int force_k, int nprod,
                bool ckpt, int nwatch, ShapeChoice* out) {
    double best_cost = -1.0;
    ShapeChoice best{0, 0, 0};
    const int wg_per_cu_needed = std::max(1, (B + num_cu - 1) / num_cu);
    int ns_forced = 0;
    if (const char* e = std::getenv("CTCFA_NS")) ns_forced = std::max(2, std::min(ctcfa::kExchangeRing / ctcfa::kGroups, std::atoi(e)));   // (a tile may run 2 NS groups ahead of its neighbour: the exchange ring has 8 slots)
    const double x = ckpt ? 0.0 : 6.0;   // extra vector instructions per cell for the decision math
    auto r512 = [](int v) { return (v + 511) / 512 * 512; };  // LDS is handed out in 512-byte units
    for (int K : kKs) {
        if (force_k && K != force_k) continue;
        if (nwatch > 0 && K > ctcfa::kWatchMaxK) continue;   // shared fills: the watch variant of the row loop
        const int U = tile_useful_cols(K);
        const int W = (Cmax + (K - 1) + U - 1) / U;   // (the left padding can take up to K-1 columns)
        const int waves_per_wg = waves_of(W, nprod);
        if (!shape_launchable(K, W, nprod)) continue;
        // Ring slots: 4 (the producer up to three blocks ahead of the tiles) unless fewer are what lets the workgroups a CU
        // needs share its LDS with backtrack workgroups of the previous batch (the pipelined schedule) -- with two of them
        // (3 slots), else with one (3, then 2 slots: 64 entries 0.279 -> 0.213 ms per pipelined step with three slots and
        // one backtrack workgroup beside, 56 entries 0.257 -> 0.218, 192 entries 0.477 -> 0.402 with two slots) -- or, where
        // no backtrack fits beside anyway, with each other: 192 entries 103 / 78 KB with four / three slots (465 -> 330 us
        // alone), 256 entries 135 / 102 / 70 KB with four / three / two (543 -> 388 us).  Where everything fits, fewer slots
        // are slower (32 entries 140 -> 163 us with two, 128 entries 298 -> 328).
        int NS = ns_forced ? ns_forced : 4;
        if (!ns_forced) {
            const int want = std::min(wg_per_cu_needed, 2);
            auto fits = [&](int ns, int nbt) {
                return want * r512(lds_bytes_fill(ns, W, K, VP, nwatch)) + nbt * r512(lds_beside) <= lds_limit;
            };
            auto groups = [&](int ns) { return lds_limit / r512(lds_bytes_fill(ns, W, K, VP, nwatch)); };
            if (fits(4, 2)) NS = 4;
            else if (fits(3, 2)) NS = 3;
            else if (fits(4, 1)) NS = 4;
            else if (fits(3, 1)) NS = 3;
            else if (fits(2, 1)) NS = 2;
            else if (groups(4) >= want) NS = 4;
            else if (groups(3) > groups(4)) NS = groups(3) >= want || groups(2) <= groups(3) ? 3 : 2;
            else if (groups(2) > groups(4)) NS = 2;
            if (lds_bytes_fill(NS, W, K, VP, nwatch) > lds_limit) NS = 3;
        }
        const int lds = lds_bytes_fill(NS, W, K, VP, nwatch);
        if (lds > lds_limit) continue;
        const int active = W + nprod;   // (padding waves leave at once)
        const int g_lds = std::max(lds_limit / r512(lds), 1);
        const int g_wave = std::max(32 / active, 1);
        const int g_vgpr = std::max(1, 4 * (512 / vgprs_of(K, ckpt)) / active);
        int G = std::max(1, std::min(g_lds, std::min(g_wave, g_vgpr)));
        const int g_eff = std::min(G, wg_per_cu_needed);
        const int rounds = (wg_per_cu_needed + G - 1) / G;
        const double wave_bound = 62.0 + (21.4 + 5.0 * x) * K;
        const double simd_bound = (g_eff * W / 4.0) * (5.0 + (15.0 + 2.5 * x) * K);
        double cost = rounds * std::max(wave_bound, simd_bound) * (1.0 + 0.03 * K);
        // pipelined schedule: two backtrack workgroups of the previous batch (4 waves, kBacktrackVgprs
        // registers, lds_beside bytes each) want to sit beside the fill on every CU
        const int act_cu = g_eff * active;
        if (act_cu + 8 > 32 || vgprs_of(K, ckpt) * ((act_cu + 3) / 4) + 2 * kBacktrackVgprs > 512 ||
            g_eff * r512(lds) + 2 * r512(lds_beside) > lds_limit)
            cost *= 1.25;
        // Many coupled tiles are fragile beside another kernel: a tile that loses its issue slots to a
        // backtrack wave holds up every tile to its right (measured: 8 tiles, decision-word mode, +44 % beside
        // the backtrack, 4 tiles +0 %).  In checkpoint mode the backtrack steps back instead (bt_low_prio).
        if (!ckpt && W > 6) cost *= 1.25;
        // Vocabularies above 64 entries: two labels in three share an LDS bank pair of the (e, m) rows, every gather
        // is a conflict -- and sixteen waves queueing at one LDS fare worse than eight (r02_vocab.txt: V = 100 as
        // 14 one-column tiles 454 us, V = 128 as 6 two-column tiles 401 us)
        if (VP > 64 && K == 1 && W > 8) cost *= 1.3;
        cost += 1e-3 * waves_per_wg;
        if (best_cost < 0.0 || cost < best_cost) {
            best_cost = cost;
            best = {K, W, NS};
        }
    }
    if (!best.K) return false;
    *out = best;
    return true;
}

}  // namespace

extern "C" {

int ctcfa_version(void) { return CTCFA_VERSION; }

int ctcfa_build_flags(void) {
    int f = 0;
#if CTCFA_ABL != 0
    f |= CTCFA_BUILD_ABLATED;
#endif
#if defined(CTCFA_STAMP) || defined(CTCFA_BT_STAMP)
    f |= CTCFA_BUILD_STAMPS;
#endif
#ifdef CTCFA_DEV_VP32_ONLY
    f |= CTCFA_BUILD_ONE_PITCH;
#endif
#if defined(CTCFA_NO_DEADZONE) || defined(CTCFA_DEBUG_SPIN) || CTCFA_PF != 2 || CTCFA_POLL_LEAD != 4 || CTCFA_BODY_BLOCKS != 2 || CTCFA_PROD_PACE != 0 || CTCFA_MASKED_PUBLISH != 1 || CTCFA_OWNER_DEFER != 0 || CTCFA_LEAN_HANDOVER != 1 || CTCFA_ADDTID_PRODUCER != 1 || CTCFA_LEAN_ALL_PITCHES != 0 || \
    CTCFA_NBR_SLEEP != 1 || CTCFA_TWO_PROD32 != 0 || CTCFA_VGPR_CAP != 1 || CTCFA_PRODUCER_PRIO != 1 || CTCFA_TILE_PRIO_BASE != 2 || CTCFA_TRACE_NT != 1 || CTCFA_SB_RING != 8 || CTCFA_SB_MARGIN != 15
    f |= CTCFA_BUILD_RETUNED;
#endif
    return f;
}

const char* ctcfa_status_string(int s) {
    switch (s) {
        case CTCFA_ST_OK: return "ok";
        case CTCFA_ST_AUDIO_SHORTER_THAN_TEXT: return "Audio is shorter than text!";
        case CTCFA_ST_BACKTRACK_FAILED: return "backtrack left the trellis (IndexError in ctc_segmentation)";
        case CTCFA_ST_WINDOWED_UNSUPPORTED: return "windowed DP regime (T > min_window_size) with T beyond the LDS column buffer (~40 000 frames)";
        case CTCFA_ST_TEXT_TOO_LONG: return "more label columns than one workgroup of the fill kernel covers";
        case CTCFA_ST_INTERNAL: return "internal error: a wave of the fill kernel gave up waiting for a progress counter";
        case CTCFA_ST_TOO_MANY_LABELS: return "the text uses more than 31 vocabulary entries beside the blank (CTCFA_FLAG_TEXTS_OF_31_LABELS promised otherwise)";
        default: return "unknown status";
    }
}

void ctcfa_default_params(ctcfa_params* p) {
    if (!p) return;
    p->blank = 0;
    p->flags = CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO;
    p->min_window_size = 8000;
    p->max_window_size = 100000;
    p->score_min_mean_over_L = 30;
    p->reserved = 0;
    p->index_duration = 0.025;
}

int ctcfa_engine_create(ctcfa_engine** out, int device) {
    if (!out) return set_err(nullptr, CTCFA_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(nullptr, CTCFA_ERR_HIP,
                       "no HIP device: this library has no CPU fallback (hipGetDeviceCount)");
    if (device < 0 || device >= n) return set_err(nullptr, CTCFA_ERR_INVALID, "device out of range");
    ctcfa_engine* eng = new ctcfa_engine();
    eng->device = device;
    DeviceGuard on_device(device);
    hipDeviceProp_t prop;
    HIP_TRY(eng, hipGetDeviceProperties(&prop, device));
    eng->num_cu = prop.multiProcessorCount;
    eng->lds_limit = (int)std::min<size_t>(prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 65536, 160 * 1024);
    if (std::string(prop.gcnArchName).find("gfx950") == std::string::npos &&
        !std::getenv("CTCFA_ALLOW_ANY_ARCH")) {
        std::string m = std::string("device arch ") + prop.gcnArchName + " is not gfx950";
        delete eng;
        return set_err(nullptr, CTCFA_ERR_UNSUPPORTED, m);
    }
    HIP_TRY(eng, hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
    if (!std::getenv("CTCFA_NO_DONE_WORD") &&
        hipMalloc(reinterpret_cast<void**>(&eng->d_done_count), 64) == hipSuccess &&
        hipMemset(eng->d_done_count, 0, 64) == hipSuccess &&
        hipHostMalloc(reinterpret_cast<void**>(&eng->h_done), 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer(reinterpret_cast<void**>(&eng->d_done_word), eng->h_done, 0) == hipSuccess) {
        *eng->h_done = 0;
    } else {   // (calls wait for their stream then, as before)
        (void)hipGetLastError();
        eng->d_done_word = nullptr;
    }
    *out = eng;
    return CTCFA_OK;
}

void ctcfa_engine_destroy(ctcfa_engine* eng) {
    if (!eng) return;
    DeviceGuard on_device(eng->device);
    if (eng->trace_calls) {
        const char* names[7] = {"plan", "pack", "enqueue uploads", "launch kernels", "enqueue download", "wait", "unpack"};
        std::fprintf(stderr, "ctcfa call trace, %lld calls, us per call:", (long long)eng->trace_calls);
        for (int k = 0; k < 7; ++k) std::fprintf(stderr, "  %s %.1f", names[k], eng->trace_ns[k] / 1e3 / (double)eng->trace_calls);
        std::fprintf(stderr, "\n");
    }
    for (auto& sl : eng->scratch)
        if (sl.p) (void)hipFree(sl.p);
    if (eng->h_in) (void)hipHostFree(eng->h_in);
    if (eng->h_out) (void)hipHostFree(eng->h_out);
    if (eng->h_done) (void)hipHostFree(eng->h_done);
    if (eng->d_done_count) (void)hipFree(eng->d_done_count);
    if (eng->stream) (void)hipStreamDestroy(eng->stream);
    delete eng;
}

const char* ctcfa_last_error(const ctcfa_engine* eng) { return eng ? eng->err.c_str() : g_err.c_str(); }

int ctcfa_max_label_columns(const ctcfa_engine* eng, int32_t vocab) {
    if (!eng || vocab <= 0) return 0;
    const bool gather = vocab > kMaxStagedVocab;
    const int VP = gather ? 128 : vocab_pitch(vocab);
    const int nprod = (!gather && (VP > 32 || vocab < 32 || CTCFA_TWO_PROD32 || CTCFA_ADDTID_PRODUCER)) ? 2 : 1;
    return label_column_limit(eng->lds_limit, VP, nprod, gather);
}

void ctcfa_plan_destroy(ctcfa_plan* plan) {
    if (!plan) return;
    DeviceGuard on_device(plan->eng ? plan->eng->device : 0);
    if (plan->scratch_owned) {
        plan->d_segs = nullptr;
        plan->d_roles = nullptr;
        plan->d_bits[0] = nullptr;
        plan->d_lastcol[0] = nullptr;
        plan->d_watch = nullptr;
        plan->d_win_list = nullptr;
        plan->d_win_table = nullptr;
        plan->d_win_offs = nullptr;
    }
    if (plan->d_watch) (void)hipFree(plan->d_watch);
    if (plan->d_segs) (void)hipFree(plan->d_segs);
    if (plan->d_roles) (void)hipFree(plan->d_roles);
    if (plan->d_win_list) (void)hipFree(plan->d_win_list);
    if (plan->d_win_table) (void)hipFree(plan->d_win_table);
    if (plan->d_win_offs) (void)hipFree(plan->d_win_offs);
    for (int q = 0; q < kWorkspaces; ++q) {
        if (plan->d_bits[q]) (void)hipFree(plan->d_bits[q]);
        if (plan->d_lastcol[q]) (void)hipFree(plan->d_lastcol[q]);
        if (plan->ev_fill_done[q]) (void)hipEventDestroy(plan->ev_fill_done[q]);
        if (plan->ev_bt_done[q]) (void)hipEventDestroy(plan->ev_bt_done[q]);
    }
    if (plan->side) (void)hipStreamDestroy(plan->side);
    for (auto& e : plan->ev)
        if (e) (void)hipEventDestroy(e);
    delete plan;
}

}  // extern "C"

namespace {

hipError_t scratch_get(ctcfa_engine* eng, int slot, void** p, size_t bytes) {
    ScratchSlot& sl = eng->scratch[slot];
    if (sl.cap < bytes) {
        if (sl.p) (void)hipFree(sl.p);
        sl.p = nullptr;
        sl.cap = 0;
        const size_t cap = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&sl.p, cap);
        if (e != hipSuccess) return e;
        e = hipMemset(sl.p, 0, cap);   // (the workspace's error word must not start out as some future run's number)
        if (e != hipSuccess) {
            (void)hipFree(sl.p);   // (cap is still 0: the next call would otherwise allocate over this pointer)
            sl.p = nullptr;
            return e;
        }
        sl.cap = cap;
    }
    *p = sl.p;
    return hipSuccess;
}

// use_scratch: the plan lives for one synchronous ctcfa_align_batch call; its device tables and
// workspace come from the engine's grow-only scratch and the uploads go onto the engine's stream
//
// emission_of (NULL: every segment has emissions of its own): emission_of[b] = e <= b, the segment whose
// emissions b uses (emission_of[e] == e, T[e] == T[b]).  Members of such a group whose label sequence is
// a proper prefix of the group's longest (host `labels`, all segments back to back; NULL: the caller
// vouches for it) share its trellis fill: SegDesc::watch_*.
int plan_create_impl(ctcfa_engine* eng, ctcfa_plan** out, const ctcfa_params* params, int32_t batch,
                     int32_t vocab, const int32_t* T, const int32_t* C, const int32_t* U,
                     int32_t force_k, bool use_scratch, const int32_t* emission_of = nullptr,
                     const int32_t* labels = nullptr, int32_t label_width = 1) {
    if (label_width < 1 || label_width > ctcfa::kMaxSpan)
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, "label_width must be in [1,16]");
    if (label_width > 1) emission_of = nullptr;   // (no shared fills for label matrices; callers pass none)
    if (!eng || !out || !params || !T || !C) return set_err(eng, CTCFA_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (batch <= 0 || vocab <= 0) return set_err(eng, CTCFA_ERR_INVALID, "batch and vocab must be positive");
    if (params->blank < 0 || params->blank >= vocab) return set_err(eng, CTCFA_ERR_INVALID, "blank outside vocabulary");
    if (!(params->index_duration > 0.0)) return set_err(eng, CTCFA_ERR_INVALID, "index_duration must be > 0");
    // blank_transition_cost_zero (gratis_blank): the fill charges nothing for staying in a column labelled
    // blank while the package's backtrack still assumes max(blank, label) -- only checkpoint mode keeps
    // the two apart (the decision words of the other mode are computed with the fill's step)
    const bool gratis = (params->flags & CTCFA_FLAG_BLANK_TRANSITION_COST_ZERO) != 0;
    static const char* const kGratisWide =
        "blank_transition_cost_zero needs a vocabulary of at most 64 entries, or a plan created with its labels whose texts use at "
        "most 62 entries each beside the blank (the host-buffer and resident entries also take larger ones whose launch looks at "
        "no more than 63 distinct labels)";
    if (gratis && vocab > 64 &&
        !((labels || (params->flags & CTCFA_FLAG_TEXTS_OF_31_LABELS)) && label_width == 1 && vocab <= kMaxStagedVocab))   // (a narrowed plan takes it: decided below)
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, kGratisWide);
    if (params->score_min_mean_over_L < 1 || params->score_min_mean_over_L > kMaxScoreLength)
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, "score_min_mean_over_L must be in [1, 1048576]");
    // wide vocabularies (sub-word models) take the gather kernel: no LDS staging of vocabulary rows
    const bool gather = vocab > kMaxStagedVocab || std::getenv("CTCFA_FORCE_GATHER") != nullptr;
    if (gather && !(params->flags & CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO))
        return set_err(eng, CTCFA_ERR_UNSUPPORTED,
                       "vocab > 256 needs preamble_transition_cost_zero (the package default)");
    if (gratis && gather)
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, "blank_transition_cost_zero is not built for the wide-vocabulary fill kernel");
    GUARD_DEVICE(eng);

    ctcfa_plan* pl = new ctcfa_plan();
    pl->eng = eng;
    pl->prm = *params;
    pl->B = batch;
    pl->V = vocab;
    pl->S = label_width;
    // LDS row pitch: the vocabulary rounded up to a compiled size.  Character vocabularies of
    // wav2vec2 models sit between 32 and 64 (the reference's Spanish model: 38 tokens), where
    // a pitch of 64 would cost a second workgroup per CU.
    pl->VP = vocab_pitch(vocab);
    pl->have_utt = (U != nullptr);

    for (int b = 0; b < batch; ++b) {
        if (T[b] < 1 || C[b] < 2 || (U && U[b] < 0)) {
            delete pl;
            return set_err(eng, CTCFA_ERR_INVALID, "need T >= 1, C >= 2 ([-1, ..., blank]) and U >= 0");
        }
    }
    if (emission_of)
        for (int b = 0; b < batch; ++b) {
            const int e = emission_of[b];
            if (e < 0 || e > b || emission_of[e] != e || T[e] != T[b]) {
                delete pl;
                return set_err(eng, CTCFA_ERR_INVALID,
                               "emission_of[b] must name an earlier segment (or b itself) that has emissions of its own and the same T");
            }
        }
    pl->gather = gather;
    // Every vocabulary but the 32-entry one (whose rows a single wave moves with four wide loads per block) takes
    // two producer waves, each staging half the rows of every block: one alone cannot keep six tiles fed.
    int nprod = (!gather && (pl->VP > 32 || vocab < 32 || CTCFA_TWO_PROD32 || CTCFA_ADDTID_PRODUCER)) ? 2 : 1;
    pl->VPbt = pl->VP;
    // What the shapes alone decide, per segment (the package's assertion and window rule): only the
    // segments that go through the fill kernel count for its launch shape -- one over-long text in a
    // batch is that segment's status, not the batch's failure.
    const int64_t lds_dyn_max = (int64_t)eng->lds_limit - 64;  // the windowed kernel also has a few static words
    const int c_limit = label_column_limit(eng->lds_limit, pl->VP, nprod, gather);   // label columns the widest fill shape can take
    std::vector<int32_t> pre(batch, CTCFA_ST_OK);
    int Cmax = 2, Tmax = 1;
    for (int b = 0; b < batch; ++b) {
        if (C[b] > T[b]) pre[b] = CTCFA_ST_AUDIO_SHORTER_THAN_TEXT;
        else if (T[b] > params->min_window_size || label_width > 1)   // label matrices: the literal (windowed) kernel, any T
            pre[b] = ((int64_t)T[b] * 4 > lds_dyn_max) ? CTCFA_ST_WINDOWED_UNSUPPORTED : ctcfa::kPreWindowed;
        else if (C[b] > c_limit) pre[b] = CTCFA_ST_TEXT_TOO_LONG;
        if (pre[b] == CTCFA_ST_OK) {
            Cmax = std::max(Cmax, (int)C[b]);
            Tmax = std::max(Tmax, (int)T[b]);
        }
    }
    // Shared fills: per emission group, the longest aligned member leads; members whose labels are a
    // proper prefix of its labels follow (everything else over the same emissions is filled by itself).
    std::vector<int32_t> leader(batch);
    for (int b = 0; b < batch; ++b) leader[b] = b;
    int nwatch_bound = 0;
    if (emission_of && !gather) {
        std::vector<int64_t> lab_at(batch);
        {
            int64_t o = 0;
            for (int b = 0; b < batch; ++b) {
                lab_at[b] = o;
                o += C[b];
            }
        }
        std::vector<int32_t> lead_of_block(batch, -1);
        for (int b = 0; b < batch; ++b) {   // longest aligned member of every emission block (first on ties)
            if (pre[b] != CTCFA_ST_OK) continue;
            int32_t& L = lead_of_block[emission_of[b]];
            if (L < 0 || C[b] > C[L]) L = b;
        }
        std::vector<int32_t> members(batch, 0);
        for (int b = 0; b < batch; ++b) {
            if (pre[b] != CTCFA_ST_OK) continue;
            const int L = lead_of_block[emission_of[b]];
            if (L == b || C[b] >= C[L]) continue;
            if (labels && std::memcmp(labels + lab_at[b], labels + lab_at[L], sizeof(int32_t) * (size_t)C[b]) != 0) continue;
            if (members[L] + 2 > ctcfa::kMaxWatch) continue;   // (the leader's own column takes one entry)
            leader[b] = L;
            ++members[L];
        }
        for (int b = 0; b < batch; ++b)
            if (members[b]) nwatch_bound = std::max(nwatch_bound, members[b] + 1);
    }
    // Narrowed plan (round 4): a vocabulary of 33 .. 256 entries whose texts -- every segment's own -- use at most 31 of them
    // beside the blank runs through the 32-ENTRY fill kernel and the 32-entry checkpoint-mode backtrack: both stage the
    // entries the segment's text uses (its "ring": ctcfa::narrow_build, derived from the labels on the device by every
    // workgroup).  A character model's window does (the reference's 38-token model; `bench.py --vocab 38 --alphabet 28`).
    // Asked for with CTCFA_FLAG_TEXTS_OF_31_LABELS (the caller's promise: a text that breaks it gets status
    // CTCFA_ST_TOO_MANY_LABELS), or decided here when the labels are at hand (the host-buffer entries;
    // ctcfa_plan_create_shared with labels).  Shared fills narrow like any other plan: the leader's ring holds every label of
    // its prefixes, and each member's backtrack derives the ring of its own text.
    bool narrowed = false;
    int ring = 32;   // entries of the narrowed plan's ring: 32 (texts of at most 31 labels), 64 (at most 62; vocabularies above 64 entries)
    if (!gather && CTCFA_ADDTID_PRODUCER && label_width == 1 && vocab > 32 && vocab <= kMaxStagedVocab &&
        !std::getenv("CTCFA_NO_NARROW")) {
        if (params->flags & CTCFA_FLAG_TEXTS_OF_31_LABELS) {
            narrowed = true;
        } else if (labels) {
            narrowed = true;
            int most = 0;   // labels beside the blank in the text that uses most
            int64_t o = 0;
            std::vector<int8_t> seen(vocab);
            for (int b = 0; b < batch && narrowed; o += C[b], ++b) {
                if (pre[b] != CTCFA_ST_OK) continue;
                std::fill(seen.begin(), seen.end(), 0);
                seen[params->blank] = 1;
                int used = 0;
                for (int c = 1; c < C[b] && narrowed; ++c) {
                    const int32_t g = labels[o + c];
                    if (g < 0 || g >= vocab) narrowed = false;   // (such a segment never gets here through the Python mirror; not ours to judge)
                    else if (!seen[g]) {
                        seen[g] = 1;
                        narrowed = ++used <= ctcfa::narrow_max_labels(64);
                    }
                }
                most = std::max(most, used);
            }
            // (a 64-entry ring is the 64-entry kernels' pace: no gain for a vocabulary that has them anyway)
            if (narrowed && most > ctcfa::narrow_max_labels(32)) {
                if (vocab > 64 && !std::getenv("CTCFA_NO_RING64")) ring = 64;
                else narrowed = false;
            }
        }
    }
    if (narrowed) {
        pl->narrow = true;
        pl->VP = ring;
        pl->VPbt = ring;
        nprod = 2;
    }
    if (gratis && vocab > 64 && !narrowed) {
        delete pl;
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, kGratisWide);
    }
    auto n_fill_of = [&]() {
        int n = 0;
        for (int b = 0; b < batch; ++b) n += (pre[b] == CTCFA_ST_OK && leader[b] == b);
        return std::max(n, 1);
    };
    // Checkpoint mode (vocab <= 64): the fill stores the table row every 32-row block ends in instead
    // of decision words (3 instead of 9 VALU per cell) and the backtrack recomputes the decisions
    // along the path -- a longer, serial backtrack.  It pays where the fill is what a batch waits
    // for: many label columns per segment, or enough segments per CU.
    // CTCFA_CHECKPOINT=1 / CTCFA_DECISION_BITS=1 force one mode (tests, tuning).
    {
        const bool can = !gather && pl->VPbt <= 64;
        // (the host-buffer entry runs fill and backtrack one after the other: there the longer
        // backtrack only pays once the fill is several times its length)
        const int64_t nf = n_fill_of();
        // Round 3 (profiles/r03_modes.txt): with the strider backtrack, checkpoint mode is ahead for every batch of
        // long segments -- 512 x 3000 frames: 128 label columns 0.106 against 0.139 ms per step, 254: 0.120 / 0.164,
        // 380: 0.142 / 0.207, 640: 0.156 / 0.253 -- and level or a little behind for short windows (4096 x 425 x 54:
        // 0.128 / 0.121; the word-level and corpus streams: +-2 %), which stay with decision words.
        // A lone window (at most one segment per CU) is one wave's dependent instruction stream per tile, 6 cycles an
        // instruction with nothing to hide them behind: 3 instead of 10 per row is worth more than the longer backtrack
        // from about 200 frames on (T = 499: 17.9 + 12.6 us against 24.3 + 12.7, T = 700: 23.8 + 11.7 against 32.4 + 14.1,
        // T = 288: 10.4 + 11.6 against 14.2 + 9.0, T = 224: 8.6 + 7.8 against 11.5 + 6.2, T = 160: 6.7 + 9.8 against
        // 8.9 + 6.6; profiles/r03_small_window_modes.txt).
        const bool lone = batch <= eng->num_cu;
        const bool pays = (lone && Tmax >= 208 && !std::getenv("CTCFA_NO_LONE_RULE")) ||
                          (use_scratch ? nf * Cmax >= 1024 * 1024
                                       : Cmax >= 544 || Tmax >= 900 || (nf * Cmax >= 250000 && Tmax >= 1000 && Cmax >= 192));
        // (57..64 entries: a strider's LDS slot is 8 KB and only ONE backtrack workgroup fits a CU beside two fill
        // workgroups, with a ring of three slots -- pick_shape: 0.213 ms per pipelined step, 0.240 with round 2's 15.8 KB
        // backtrack, 0.272 with decision words)
        pl->ckpt = can && (std::getenv("CTCFA_CHECKPOINT") ? true : pays) && !std::getenv("CTCFA_DECISION_BITS");
        if (gratis) pl->ckpt = true;   // (vocab <= 64 or a narrowed plan: checked above)
        if (pl->narrow) pl->ckpt = true;   // (the other backtrack does not know the ring: it could not tell a text that breaks the promise)
    }
    // what one backtrack workgroup of this batch will ask for (the exact figure is set further down)
    int bt_lds_estimate;
    {
        const int Tb = Tmax;
        const int rec = ((Tb + ctcfa::kRows - 1) / ctcfa::kRows * 8 + 15) / 16 * 16;
        bt_lds_estimate = pl->ckpt ? lds_bytes_strider(rec, (Cmax + 15) / 16 * 16, pl->VPbt, strider_waves(batch <= eng->num_cu), Tb, pl->narrow) : rec + Tb * 4;
    }
    ShapeChoice shape{0, 0, 0};
    if (gather) {
        const int W = std::max(1, (Cmax - 1 + 63) / 64);  // label columns 1 .. C-1, one per lane
        shape = {1, W, 0};
        pl->roles = ctcfa::FillRoles{};
        pl->roles.nwaves = W;   // W compute waves, no producer
        pl->roles.nstages = W;
        pl->roles.cpad = 64 * W;
    } else {
        bool ok = pick_shape(n_fill_of(), Cmax, pl->VP, eng->lds_limit, bt_lds_estimate, eng->num_cu, force_k, nprod,
                             pl->ckpt, nwatch_bound, &shape);
        if (!ok && nwatch_bound > 0) {   // too many label columns for the tile widths that can watch: no sharing
            for (int b = 0; b < batch; ++b) leader[b] = b;
            nwatch_bound = 0;
            ok = pick_shape(n_fill_of(), Cmax, pl->VP, eng->lds_limit, bt_lds_estimate, eng->num_cu, force_k, nprod,
                            pl->ckpt, 0, &shape);
        }
        if (!ok) {
            delete pl;
            return set_err(eng, force_k ? CTCFA_ERR_INVALID : CTCFA_ERR_UNSUPPORTED,
                           force_k ? "cols_per_lane: not a compiled tile width, or too narrow for this batch"
                                   : "no launch shape fits");
        }
        pl->roles = tile_roles(shape.K, shape.W, shape.NS, nprod);
    }
    pl->K = shape.K;
    pl->W = shape.W;
    if (!gather) {
        pl->fill_fn = select_fill(pl->K, pl->VP, pl->ckpt);
        if (!pl->fill_fn) {
            delete pl;
            return set_err(eng, CTCFA_ERR_INVALID, "unsupported cols_per_lane");
        }
    }
    const int K = pl->K;
    const int64_t Cpad = pl->roles.cpad;
    const int tileU = gather ? 64 : tile_useful_cols(K);
    const int tileHL = gather ? 0 : ctcfa::halo_lanes(K);
    // Watch columns: one lane publishes one column.  A member whose last column shares a lane with
    // another member's (utterances of one or two labels under wide tiles) is filled by itself.
    std::vector<std::vector<int32_t>> followers(batch);
    if (nwatch_bound > 0) {
        auto lane_key = [&](int pcol) { return (pcol / tileU) * 64 + tileHL + (pcol % tileU) / K; };
        std::vector<std::vector<int32_t>> taken(batch);
        for (int b = 0; b < batch; ++b) {
            const int L = leader[b];
            if (L == b) continue;
            const int shiftL = ((K - 1) - (C[L] - 1) % K + K) % K;
            if (taken[L].empty()) taken[L].push_back(lane_key(C[L] - 1 + shiftL));
            const int key = lane_key(C[b] - 1 + shiftL);
            if (std::find(taken[L].begin(), taken[L].end(), key) != taken[L].end()) {
                leader[b] = b;
                continue;
            }
            taken[L].push_back(key);
            followers[L].push_back(b);
        }
        nwatch_bound = 0;
        for (int b = 0; b < batch; ++b)
            if (!followers[b].empty()) nwatch_bound = std::max(nwatch_bound, (int)followers[b].size() + 1);
    }
    pl->nwatch_max = nwatch_bound;
    pl->roles.reserved[0] = nwatch_bound;
    pl->lds_fill = gather ? (pl->W + 1) * ctcfa::kBndPitch * 4 + 64 * 4 + ctcfa::kSinkBytes
                          : lds_bytes_fill(shape.NS, pl->W, K, pl->VP, nwatch_bound);
    pl->n_fill = n_fill_of();
    pl->segs.resize(batch);
    int64_t lpz_off = 0, lab_off = 0, frm_off = 0, utt_off = 0, bits_off = 0;
    for (int b = 0; b < batch; ++b) {
        SegDesc& s = pl->segs[b];
        const bool own_emissions = !emission_of || emission_of[b] == b;
        s.lpz_off = own_emissions ? lpz_off : pl->segs[emission_of[b]].lpz_off;
        s.lab_off = lab_off;
        s.frm_off = frm_off;
        s.utt_off = utt_off;
        s.bits_off = bits_off;
        s.T = T[b];
        s.C = C[b];
        s.U = U ? U[b] : 0;
        if (gather) {  // padded column pc holds label column pc + 1 (column 0 is not a DP column)
            s.shift = -1;
            s.owner_stage = (C[b] - 2) / 64;
            s.owner_lane = (C[b] - 2) % 64;
        } else {   // left padding: the last label column must sit at k == K-1 of its lane
            s.shift = ((K - 1) - (C[b] - 1) % K + K) % K;
            const int pcl = C[b] - 1 + s.shift;   // padded column of the last label column
            s.owner_stage = pcl / tileU;
            s.owner_lane = tileHL + (pcl % tileU) / K;
        }
        s.seg_index = b;
        s.prestatus = pre[b];
        {
            if (s.prestatus == ctcfa::kPreWindowed) {
                // windowed regime: own kernel, needs T floats of LDS and a T x C fp32 table in HBM
                s.win_off = pl->win_table_floats;
                s.wcol_off = pl->win_cols;
                pl->win_table_floats += (int64_t)T[b] * C[b];
                pl->win_cols += C[b] + 2;   // (+ band_fill_kernel's two control words)
                pl->win_cmax = std::max(pl->win_cmax, (int)C[b]);
                // T floats at least (char_probs for the scoring); room for two columns of the first
                // window, or of the whole trellis when that still fits, lets the fill double-buffer
                {
                    const int64_t W0 = std::min<int64_t>(T[b], params->min_window_size);
                    int64_t want = std::max<int64_t>((int64_t)T[b] * 4, W0 * 8);
                    if ((int64_t)T[b] * 8 <= lds_dyn_max) want = (int64_t)T[b] * 8;
                    want = std::min<int64_t>(want, lds_dyn_max);
                    pl->lds_win = std::max(pl->lds_win, (int)want);
                }
                pl->win_list.push_back(b);
            }
        }
        const int nblk = (T[b] - 1 + ctcfa::kRows - 1) / ctcfa::kRows;
        pl->nblk_max = std::max(pl->nblk_max, nblk);
        if (own_emissions) {
            lpz_off += (int64_t)T[b] * vocab;
            pl->total_lpz_T += T[b];
        }
        lab_off += C[b];
        frm_off += T[b];
        utt_off += s.U;
        const bool own_fill = s.prestatus == CTCFA_ST_OK && leader[b] == b;
        if (own_fill) bits_off += (int64_t)nblk * Cpad;
        // 4TV + TC/8 + 4T + 8C + 4T (emissions and trace words once per shared fill)
        pl->alg_bytes += (own_emissions ? 4LL * T[b] * vocab : 0) + (own_fill ? (int64_t)T[b] * C[b] / 8 : 0) +
                         4LL * T[b] + 8LL * C[b] + 4LL * T[b];
    }
    for (int b = 0; b < batch; ++b) {   // followers walk their leader's trace; leaders list the watch columns
        SegDesc& s = pl->segs[b];
        if (leader[b] != b) {
            const SegDesc& l = pl->segs[leader[b]];
            s.bits_off = l.bits_off;
            s.shift = l.shift;
            s.fill_skip = 1;
        } else if (!followers[b].empty()) {
            s.watch_first = (int32_t)pl->watch.size();
            s.watch_n = (int32_t)followers[b].size() + 1;
            std::vector<int32_t> m = followers[b];
            m.push_back(b);
            std::sort(m.begin(), m.end(), [&](int x, int y) { return C[x] < C[y]; });
            for (int x : m) pl->watch.push_back(ctcfa::WatchDesc{pl->segs[x].frm_off, C[x] - 1 + s.shift, 0});
        }
    }
    pl->total_T = frm_off;
    pl->total_C = lab_off;
    pl->total_U = utt_off;
    pl->bits_words = bits_off;
    pl->rec_bytes = (std::max(1, pl->nblk_max) * 8 + 15) / 16 * 16;
    {   // the backtrack kernel keeps char_probs of its segment in LDS; windowed segments have their own kernel
        int Tbt = 1;
        for (int b = 0; b < batch; ++b)
            if (pl->segs[b].prestatus == CTCFA_ST_OK) Tbt = std::max(Tbt, (int)T[b]);
        pl->lds_bt = pl->rec_bytes + Tbt * 4;
        if (pl->ckpt) {  // + label copy; the emission ring of the walk shares its LDS with char_probs
            int Cbt = 1;
            for (int b = 0; b < batch; ++b)
                if (pl->segs[b].prestatus == CTCFA_ST_OK) Cbt = std::max(Cbt, (int)C[b]);
            pl->lab_bytes = (Cbt + 15) / 16 * 16;   // one byte per label
            pl->bt_waves = strider_waves(batch <= eng->num_cu);
            pl->bt_scorers = (pl->have_utt && !std::getenv("CTCFA_SB_NO_SCORER")) ? 1 : 0;
            pl->fol_bytes = 4 * pl->lab_bytes;
            pl->lds_bt = lds_bytes_strider(pl->rec_bytes, pl->lab_bytes, pl->VPbt, pl->bt_waves, Tbt, pl->narrow);
            while (pl->lds_bt > eng->lds_limit && pl->bt_waves > 1)   // (a very long lone segment: fewer slots rather than no plan)
                pl->lds_bt = lds_bytes_strider(pl->rec_bytes, pl->lab_bytes, pl->VPbt, --pl->bt_waves, Tbt, pl->narrow);
        }
    }
    if (pl->lds_bt > eng->lds_limit) {
        delete pl;
        return set_err(eng, CTCFA_ERR_UNSUPPORTED, "segment too long for the backtrack record buffer");
    }
    pl->n_emission_blocks = 0;
    for (int b = 0; b < batch; ++b) pl->n_emission_blocks += (!emission_of || emission_of[b] == b);
    // Workgroup b of both kernels takes table entry b: longest first.  A ragged batch larger than what the
    // chip holds at once then ends with its short segments (no long one left over for the tail), and in a
    // batch that is resident at once the long segments spread over the CUs before the short ones join
    // them.  Everything a kernel needs is inside the entry (seg_index = the caller's position).
    if (batch > 1 && !std::getenv("CTCFA_NO_SORT")) {
        std::vector<int32_t> pos(batch);
        std::vector<SegDesc> sorted(pl->segs);
        std::stable_sort(sorted.begin(), sorted.end(), [](const SegDesc& a, const SegDesc& b) {
            const int64_t ca = (a.prestatus == CTCFA_ST_OK) ? (int64_t)a.T * (a.fill_skip ? 1 : a.C) : 0;
            const int64_t cb = (b.prestatus == CTCFA_ST_OK) ? (int64_t)b.T * (b.fill_skip ? 1 : b.C) : 0;
            return ca > cb;
        });
        for (int i = 0; i < batch; ++i) pos[sorted[i].seg_index] = i;
        for (auto& w : pl->win_list) w = pos[w];   // (windowed_kernel indexes the table through this list)
        pl->segs.swap(sorted);
    }

#define PLAN_TRY(expr)                                                                        \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess) {                                                               \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);                \
            ctcfa_plan_destroy(pl);                                                           \
            return set_err(eng, _e == hipErrorOutOfMemory ? CTCFA_ERR_NOMEM : CTCFA_ERR_HIP, m); \
        }                                                                                     \
    } while (0)
    if (use_scratch) {
        // transient plan of a host-buffer call: workspace from the engine's grow-only scratch; the role and
        // segment tables travel in the caller's packed upload (align_impl sets d_roles / d_segs)
        pl->scratch_owned = true;
        PLAN_TRY(scratch_get(eng, kSlotBits, reinterpret_cast<void**>(&pl->d_bits[0]),
                             sizeof(uint32_t) * (size_t)std::max<int64_t>(1, pl->bits_words)));
        PLAN_TRY(scratch_get(eng, kSlotLastcol, reinterpret_cast<void**>(&pl->d_lastcol[0]),
                             sizeof(float) * (size_t)(std::max<int64_t>(1, pl->total_T) + 4)));   // (+ the error word)
    } else {
        PLAN_TRY(hipMalloc(&pl->d_roles, sizeof(ctcfa::FillRoles)));
        PLAN_TRY(hipMemcpy(pl->d_roles, &pl->roles, sizeof(ctcfa::FillRoles), hipMemcpyHostToDevice));
        PLAN_TRY(hipMalloc(&pl->d_segs, sizeof(SegDesc) * (size_t)batch));
        PLAN_TRY(hipMemcpy(pl->d_segs, pl->segs.data(), sizeof(SegDesc) * (size_t)batch, hipMemcpyHostToDevice));
        PLAN_TRY(hipMalloc(&pl->d_bits[0], sizeof(uint32_t) * (size_t)std::max<int64_t>(1, pl->bits_words)));
        PLAN_TRY(hipMalloc(&pl->d_lastcol[0], sizeof(float) * (size_t)(std::max<int64_t>(1, pl->total_T) + 4)));
        PLAN_TRY(hipMemset(pl->d_lastcol[0] + std::max<int64_t>(1, pl->total_T), 0, 16));   // the error word
        if (!pl->watch.empty()) {
            PLAN_TRY(hipMalloc(&pl->d_watch, sizeof(ctcfa::WatchDesc) * pl->watch.size()));
            PLAN_TRY(hipMemcpy(pl->d_watch, pl->watch.data(), sizeof(ctcfa::WatchDesc) * pl->watch.size(), hipMemcpyHostToDevice));
        }
    }
    {   // The pipelined entry runs the backtrack of run k beside the fill of run k+1 -- where one fits beside the other.
        // Where not a single backtrack workgroup finds LDS (or registers: the 256-entry kernels take 128, two workgroups
        // fill the register file) next to the fill workgroups a CU holds, the two kernels only get in each other's way and
        // the entry runs them one after the other instead.  Measured, config 3's shape, ms per step pipelined | serial:
        // 56 entries 0.257 | 0.249, 64: 0.279 | 0.265, 192: 0.477 | 0.424, 256: 0.563 | 0.491 -- against 38: 0.177 | 0.244,
        // 48: 0.201 | 0.244, 76: 0.287 | 0.362, 128: 0.328 | 0.397, 160: 0.374 | 0.416 where one fits.
        auto r512 = [](int v) { return (v + 511) / 512 * 512; };
        const int fill_per_cu = std::min(2, std::max(1, (n_fill_of() + eng->num_cu - 1) / eng->num_cu));
        pl->overlap_ok = gather || (fill_per_cu * r512(pl->lds_fill) + r512(pl->lds_bt) <= eng->lds_limit && pl->VP < 256);
        if (std::getenv("CTCFA_ALWAYS_OVERLAP")) pl->overlap_ok = true;
    }
    if (pl->lds_fill > 48 * 1024 && pl->fill_fn)
        PLAN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(pl->fill_fn),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, pl->lds_fill));
    if (pl->lds_bt > 48 * 1024)
        PLAN_TRY(hipFuncSetAttribute(!pl->ckpt ? reinterpret_cast<const void*>(ctcfa::backtrack_kernel)
                                                : reinterpret_cast<const void*>(select_strider(pl->VPbt, pl->narrow)),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, pl->lds_bt));
    if (!pl->win_list.empty()) {
        if (use_scratch) {
            // (a plan that lives for one host-buffer call: the windowed kernel's table -- 47 MB for ONE 9 500-frame window,
            // 12 GB for 256 -- comes from the engine's grow-only scratch like the other workspaces; a hipMalloc / hipFree
            // pair of that size per call cost up to 600 ms, profiles/r03_windowed.txt.  The kernel initialises its table.)
            PLAN_TRY(scratch_get(eng, kSlotWinList, reinterpret_cast<void**>(&pl->d_win_list), sizeof(int32_t) * pl->win_list.size()));
            PLAN_TRY(scratch_get(eng, kSlotWinTable, reinterpret_cast<void**>(&pl->d_win_table), sizeof(float) * (size_t)pl->win_table_floats));
            PLAN_TRY(scratch_get(eng, kSlotWinOffs, reinterpret_cast<void**>(&pl->d_win_offs), sizeof(int32_t) * (size_t)pl->win_cols));
        } else {
            PLAN_TRY(hipMalloc(&pl->d_win_list, sizeof(int32_t) * pl->win_list.size()));
            PLAN_TRY(hipMalloc(&pl->d_win_table, sizeof(float) * (size_t)pl->win_table_floats));
            PLAN_TRY(hipMalloc(&pl->d_win_offs, sizeof(int32_t) * (size_t)pl->win_cols));
        }
        PLAN_TRY(hipMemcpy(pl->d_win_list, pl->win_list.data(), sizeof(int32_t) * pl->win_list.size(),
                           hipMemcpyHostToDevice));
        if (pl->lds_win > 48 * 1024)
            PLAN_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(ctcfa::windowed_kernel),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, pl->lds_win));
    }
#undef PLAN_TRY
    *out = pl;
    return CTCFA_OK;
}

}  // namespace

extern "C" {

int ctcfa_plan_create(ctcfa_engine* eng, ctcfa_plan** out, const ctcfa_params* params, int32_t batch,
                      int32_t vocab, const int32_t* T, const int32_t* C, const int32_t* U,
                      int32_t force_k) {
    return plan_create_impl(eng, out, params, batch, vocab, T, C, U, force_k, false);
}

int ctcfa_plan_get_info(const ctcfa_plan* pl, ctcfa_plan_info* info) {
    if (!pl || !info) return CTCFA_ERR_INVALID;
    info->batch = pl->B;
    info->cols_per_lane = pl->K;
    info->waves_per_seg = pl->roles.nstages;
    info->vocab_pitch = pl->VP > 80 ? pl->VP + 4 : pl->VP + ctcfa::kPitchPad;   // entries per ring row (4 B each above 80 entries, else 8 B)
    info->lds_bytes = pl->lds_fill;
    info->n_blocks_max = pl->nblk_max;
    info->workspace_bytes = pl->bits_words * 4 + pl->total_T * 4;
    info->algorithmic_bytes = pl->alg_bytes;
    info->total_frames = pl->total_T;
    return CTCFA_OK;
}

int ctcfa_plan_set_timing_stride(ctcfa_plan* pl, int stride) {
    if (!pl || stride < 1) return CTCFA_ERR_INVALID;
    pl->ev_stride = stride;
    pl->run_counter = 0;
    return CTCFA_OK;
}

int ctcfa_plan_set_timing(ctcfa_plan* pl, int slots) {
    if (!pl || slots < 0 || slots > 4096) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    GUARD_DEVICE(eng);   // (events belong to the device that is current when they are created)
    if (slots > 0 && slots < 2 * kWorkspaces) slots = 2 * kWorkspaces;  // the pipelined entry looks at the event of run k - kWorkspaces
    if (pl->side) {
        // a pending hand-over may be one of the timing events about to be destroyed
        HIP_TRY(eng, hipStreamSynchronize(pl->side));
        for (int q = 0; q < kWorkspaces; ++q) pl->bt_pending[q] = false;
    }
    if (slots > 0 && slots == pl->ev_slots) {   // the same events again, from the first slot (they have been recorded before:
        pl->ev_runs = 0;                        // a timed region that follows a warm-up pays no first-use cost of the runtime's)
        return CTCFA_OK;
    }
    for (auto& e : pl->ev)
        if (e) (void)hipEventDestroy(e);
    pl->ev.assign((size_t)slots * 4, nullptr);
    pl->ev_slots = slots;
    pl->ev_runs = 0;
    for (auto& e : pl->ev) HIP_TRY(eng, hipEventCreate(&e));
    return CTCFA_OK;
}

namespace {

struct RunArgs {
    const float* d_lpz;
    const int32_t* d_labels;
    const int32_t* d_utt_begin;
    int32_t* d_fol;
    float* d_char_prob;
    int32_t* d_state;
    double *d_seg_start, *d_seg_end, *d_seg_score;
    int32_t *d_t_end, *d_status;
};

int check_args(ctcfa_plan* pl, const RunArgs& a, bool* want_seg) {
    ctcfa_engine* eng = pl->eng;
    if (!a.d_lpz || !a.d_labels || !a.d_fol || !a.d_char_prob || !a.d_t_end || !a.d_status)
        return set_err(eng, CTCFA_ERR_INVALID, "NULL device buffer");
    *want_seg = a.d_utt_begin && a.d_seg_start && a.d_seg_end && a.d_seg_score;
    if (*want_seg && !pl->have_utt) return set_err(eng, CTCFA_ERR_INVALID, "plan was created without U[]");
    return CTCFA_OK;
}

// start / stop: events carried by the kernel's own dispatch packet (hipExtLaunchKernel) -- timing and
// cross-stream hand-over without separate event-record packets between two kernels
int launch_fill(ctcfa_plan* pl, const RunArgs& a, int ws, hipStream_t st, hipEvent_t start = nullptr,
                hipEvent_t stop = nullptr) {
    if (pl->gather) {
        hipExtLaunchKernelGGL(ctcfa::fill_gather_kernel, dim3(pl->B), dim3(64 * pl->W), pl->lds_fill, st, start, stop,
                              0, pl->d_segs, a.d_lpz, a.d_labels, pl->d_bits[ws], pl->d_lastcol[ws], pl->V,
                              pl->prm.blank, pl->roles.cpad);
        HIP_TRY(pl->eng, hipGetLastError());
        return CTCFA_OK;
    }
#ifdef CTCFA_STAMP   // tuning builds: the tiles' cycle stamps land in the caller's char_prob buffer (tools/stamps2.py)
    float* lastcol_arg = a.d_char_prob;
#else
    float* lastcol_arg = pl->d_lastcol[ws];
#endif
    ctcfa_engine* eng = pl->eng;
    eng->run_seq = eng->run_seq == 0x7fffffff ? 1 : eng->run_seq + 1;
    pl->last_run[ws] = eng->run_seq;
    hipExtLaunchKernelGGL(pl->fill_fn, dim3(pl->B), dim3(64 * pl->roles.nwaves), pl->lds_fill, st, start, stop, 0,
                          pl->d_segs, a.d_lpz, a.d_labels, pl->d_bits[ws], lastcol_arg, pl->V, pl->prm.blank,
                          ((pl->prm.flags & CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO) ? 1 : 0) |
                              ((pl->prm.flags & CTCFA_FLAG_BLANK_TRANSITION_COST_ZERO) ? 2 : 0) | (pl->narrow ? 4 : 0),
                          (const ctcfa::FillRoles*)pl->d_roles, (const ctcfa::WatchDesc*)pl->d_watch,
#if defined(CTCFA_STAMP) && CTCFA_STAMP == 4   // the timeline of tools/trace4.py: behind the caller's char_prob buffer (which the tool makes long enough)
                          reinterpret_cast<int32_t*>(a.d_char_prob + (pl->total_T + 1) / 2 * 2),
#else
                          reinterpret_cast<int32_t*>(pl->d_lastcol[ws] + std::max<int64_t>(1, pl->total_T)),
#endif
                          (int)pl->last_run[ws]);
    HIP_TRY(pl->eng, hipGetLastError());
    return CTCFA_OK;
}

int launch_backtrack(ctcfa_plan* pl, const RunArgs& a, bool want_seg, int ws, hipStream_t st,
                     hipEvent_t start = nullptr, hipEvent_t stop = nullptr, bool beside_fill = false) {
#ifdef CTCFA_STAMP
    return CTCFA_OK;   // (the stamps sit where the backtrack would write)
#endif
    const bool windowed = !pl->win_list.empty();
    BtParams bp;
    bp.V = pl->V;
    bp.blank = pl->prm.blank;
    bp.Cpad = pl->roles.cpad;
    bp.flags = pl->prm.flags;
    // Eight or more coupled fill tiles per segment: the (heavier, checkpoint-mode) backtrack of the previous
    // batch runs below them instead of above (512 x 1242: 0.362 -> 0.313 ms per step; with six tiles or
    // fewer the backtrack is what a step waits for and keeps its priority)
    if (pl->ckpt && pl->W >= 8) bp.flags |= ctcfa::kBtFlagLowPriority;
    // (the backtrack kernels sum at most 128 frames the NumPy way; a longer scoring window is scored again below)
    const bool rescore = want_seg && pl->prm.score_min_mean_over_L > 128;
    bp.L = std::min(pl->prm.score_min_mean_over_L, 128);
    bp.rec_bytes = pl->rec_bytes;
    bp.lab_bytes = pl->lab_bytes;
    bp.fol_bytes = pl->fol_bytes;
    bp.scorers = pl->ckpt ? pl->bt_scorers : 0;
    // Beside the fill of the next batch the striders run below the fill's tiles (config 3: 0.174 -> 0.165 ms per
    // step; a tile that loses an issue slot holds up every tile to its right), alone at the top.
    // (round 4: with the 32-entry fill 12 % faster the step was bound by the striders at priority 0 -- 0.1683 ms against
    // 0.1571 at priority 1, the producers' level; vocabulary 29: 0.1799 / 0.1655; 38 and 64: +- 1 %; profiles/r04_strider_knobs.txt)
    bp.prio = beside_fill ? 1 : 3;
    if (const char* e = std::getenv("CTCFA_SB_PRIO")) bp.prio = std::max(0, std::min(3, std::atoi(e)));
    bp.windows = 2;
    if (const char* e = std::getenv("CTCFA_SB_WINDOWS")) bp.windows = std::max(1, std::min(3, std::atoi(e)));
    bp.dur = pl->prm.index_duration;
    const ctcfa::BtArgs ba{pl->d_segs, a.d_lpz, a.d_labels, want_seg ? a.d_utt_begin : nullptr, pl->d_bits[ws],
                           pl->d_lastcol[ws],
                           pl->gather ? nullptr : reinterpret_cast<const int32_t*>(pl->d_lastcol[ws] + std::max<int64_t>(1, pl->total_T)),
                           pl->last_run[ws], bp, a.d_fol, a.d_char_prob, a.d_state, a.d_seg_start, a.d_seg_end,
                           want_seg ? a.d_seg_score : nullptr, a.d_t_end, a.d_status,
                           pl->done_value ? pl->eng->d_done_count : nullptr, pl->done_target,
                           pl->done_value ? pl->eng->d_done_word : nullptr, pl->done_value};
    if (!pl->ckpt)
        hipExtLaunchKernelGGL(ctcfa::backtrack_kernel, dim3(pl->B), dim3(ctcfa::kBtThreads), pl->lds_bt, st,
                              start, (windowed || rescore) ? nullptr : stop, 0, ba);
    else {
        // (the LDS is sized for bt_waves; beside the next batch's fill even a lone launch keeps to three striders:
        // 64 / 128 / 256 segments pipelined 0.1161 / 0.1167 / 0.1178 ms per step with three, 0.1165 / 0.1178 / 0.1190 with seven)
        const int striders = (beside_fill && !std::getenv("CTCFA_SB_WAVES")) ? std::min(pl->bt_waves, 3) : pl->bt_waves;
        hipExtLaunchKernelGGL(select_strider(pl->VPbt, pl->narrow), dim3(pl->B), dim3(64 * (striders + pl->bt_scorers)), pl->lds_bt, st,
                              start, (windowed || rescore) ? nullptr : stop, 0, ba);
    }
    HIP_TRY(pl->eng, hipGetLastError());
    if (windowed) {
        ctcfa::WinParams wp;
        wp.V = pl->V;
        wp.blank = pl->prm.blank;
        wp.flags = pl->prm.flags;
        wp.L = std::min(pl->prm.score_min_mean_over_L, 128);
        wp.min_window = pl->prm.min_window_size;
        wp.max_window = pl->prm.max_window_size;
        wp.lds_bytes = pl->lds_win;
        wp.S = pl->S;
        wp.dur = pl->prm.index_duration;
        wp.fast_walk = std::getenv("CTCFA_NO_FAST_WALK") ? 0 : 1;
        // single labels: the table by rows first (band_fill_kernel), the walk and the scoring -- and whatever that fill
        // gives up on -- in windowed_kernel.  CTCFA_NO_BAND_FILL=1: the literal fill only.
        // (vocabularies of at most 256 entries: a barrier every 16 rows -- band_fill_halo_kernel; CTCFA_BAND_ROW_BARRIER=1 or a
        // wider vocabulary: a barrier a row)
        int hk = 0;
        for (int k : {1, 2, 4})   // (eight columns per lane spill in the 16-row block loop: 144 ms against 132 for 25 000 x 6 002)
            if (!hk && (pl->win_cmax + ctcfa::band_halo_own_cols(k) - 1) / ctcfa::band_halo_own_cols(k) <= 16) hk = k;
        const size_t halo_lds = (size_t)ctcfa::band_halo_lds_bytes(pl->win_cmax, pl->V);
        const bool halo = hk != 0 && pl->V <= 256 && halo_lds <= (size_t)pl->eng->lds_limit && !std::getenv("CTCFA_BAND_ROW_BARRIER");
        const int bk = halo ? hk : pl->win_cmax <= 1024 ? 1 : pl->win_cmax <= 2048 ? 2 : pl->win_cmax <= 4096 ? 4 : pl->win_cmax <= 8192 ? 8 : 0;
        const size_t band_lds = halo ? halo_lds : (size_t)ctcfa::band_lds_bytes(pl->win_cmax, pl->V);
        wp.prefill = (pl->S == 1 && bk != 0 && pl->V <= ctcfa::kBandThreads && band_lds <= (size_t)pl->eng->lds_limit &&
                      !std::getenv("CTCFA_NO_BAND_FILL")) ? 1 : 0;
        if (wp.prefill) {
            auto fn = halo ? (bk == 1 ? ctcfa::band_fill_halo_kernel<1> : bk == 2 ? ctcfa::band_fill_halo_kernel<2>
                                       : bk == 4 ? ctcfa::band_fill_halo_kernel<4> : ctcfa::band_fill_halo_kernel<8>)
                           : (bk == 1 ? ctcfa::band_fill_kernel<1> : bk == 2 ? ctcfa::band_fill_kernel<2>
                                       : bk == 4 ? ctcfa::band_fill_kernel<4> : ctcfa::band_fill_kernel<8>);
            if (band_lds > 48 * 1024)
                HIP_TRY(pl->eng, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)band_lds));
            // (whole waves for the widest window's columns, and at least V lanes: they stage the emission rows)
            const int per_wave = halo ? ctcfa::band_halo_own_cols(bk) : 64 * bk;
            const int band_threads = std::min(ctcfa::kBandThreads, std::max((pl->win_cmax + per_wave - 1) / per_wave * 64, (pl->V + 63) / 64 * 64));
            hipExtLaunchKernelGGL(fn, dim3((unsigned)pl->win_list.size()), dim3(band_threads), band_lds, st, nullptr, nullptr, 0,
                                  (const SegDesc*)pl->d_segs, (const int32_t*)pl->d_win_list, a.d_lpz, a.d_labels, pl->d_win_table,
                                  pl->d_win_offs, wp);
            HIP_TRY(pl->eng, hipGetLastError());
        }
        hipExtLaunchKernelGGL(ctcfa::windowed_kernel, dim3((unsigned)pl->win_list.size()), dim3(ctcfa::kWinThreads),
                              pl->lds_win, st, nullptr, rescore ? nullptr : stop, 0, (const SegDesc*)pl->d_segs,
                              (const int32_t*)pl->d_win_list, a.d_lpz, a.d_labels,
                              want_seg ? a.d_utt_begin : (const int32_t*)nullptr, pl->d_win_table, pl->d_win_offs, wp,
                              a.d_fol, a.d_char_prob, a.d_state, a.d_seg_start, a.d_seg_end,
                              want_seg ? a.d_seg_score : (double*)nullptr, a.d_t_end, a.d_status);
        HIP_TRY(pl->eng, hipGetLastError());
    }
    if (rescore) {
        hipExtLaunchKernelGGL(ctcfa::rescore_kernel, dim3(pl->B), dim3(ctcfa::kRescoreThreads), 0, st, nullptr, stop, 0,
                              (const SegDesc*)pl->d_segs, a.d_utt_begin, (const int32_t*)a.d_fol, (const float*)a.d_char_prob,
                              pl->prm.score_min_mean_over_L, pl->prm.index_duration, a.d_seg_start, a.d_seg_end, a.d_seg_score,
                              (const int32_t*)a.d_status);
        HIP_TRY(pl->eng, hipGetLastError());
    }
    return CTCFA_OK;
}

}  // namespace

int ctcfa_plan_run_device(ctcfa_plan* pl, const float* d_lpz, const int32_t* d_labels,
                          const int32_t* d_utt_begin, int32_t* d_fol, float* d_char_prob,
                          int32_t* d_state, double* d_seg_start, double* d_seg_end,
                          double* d_seg_score, int32_t* d_t_end, int32_t* d_status, void* stream) {
    if (!pl) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    const RunArgs a{d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state,
                    d_seg_start, d_seg_end, d_seg_score, d_t_end, d_status};
    bool want_seg = false;
    int rc = check_args(pl, a, &want_seg);
    if (rc != CTCFA_OK) return rc;
    GUARD_DEVICE(eng);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // NULL = the default (null) stream
    // a pipelined run may still be reading workspace 0 on the side stream
    // (bt_pending stays set: the wait orders THIS stream only; a later pipelined run on another stream still has to see it)
    if (pl->bt_pending[0]) {
        if (hipEventQuery(pl->bt_done_ev[0]) == hipSuccess) pl->bt_pending[0] = false;   // finished: nothing to order
        else HIP_TRY(eng, hipStreamWaitEvent(st, pl->bt_done_ev[0], 0));
        (void)hipGetLastError();   // (a "not ready" from the query is no error)
    }
    const bool timed = pl->ev_slots && (pl->run_counter++ % pl->ev_stride == 0);
    hipEvent_t* ev = timed ? &pl->ev[(size_t)(pl->ev_runs % pl->ev_slots) * 4] : nullptr;
    if ((rc = launch_fill(pl, a, 0, st, ev ? ev[0] : nullptr, ev ? ev[1] : nullptr)) != CTCFA_OK) return rc;
    if ((rc = launch_backtrack(pl, a, want_seg, 0, st, ev ? ev[2] : nullptr, ev ? ev[3] : nullptr)) != CTCFA_OK) return rc;
    if (ev) pl->ev_runs++;
    return CTCFA_OK;
}

int ctcfa_plan_run_pipelined(ctcfa_plan* pl, const float* d_lpz, const int32_t* d_labels,
                             const int32_t* d_utt_begin, int32_t* d_fol, float* d_char_prob,
                             int32_t* d_state, double* d_seg_start, double* d_seg_end,
                             double* d_seg_score, int32_t* d_t_end, int32_t* d_status, void* stream) {
    if (!pl) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    const RunArgs a{d_lpz, d_labels, d_utt_begin, d_fol, d_char_prob, d_state,
                    d_seg_start, d_seg_end, d_seg_score, d_t_end, d_status};
    bool want_seg = false;
    int rc = check_args(pl, a, &want_seg);
    if (rc != CTCFA_OK) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // NULL = the default (null) stream
    GUARD_DEVICE(eng);
    if (!pl->side) {  // first use: the other workspaces, side stream, hand-over events
        for (int w = 1; w < kWorkspaces; ++w) {
            HIP_TRY(eng, hipMalloc(&pl->d_bits[w], sizeof(uint32_t) * (size_t)std::max<int64_t>(1, pl->bits_words)));
            HIP_TRY(eng, hipMalloc(&pl->d_lastcol[w], sizeof(float) * (size_t)(std::max<int64_t>(1, pl->total_T) + 4)));
            HIP_TRY(eng, hipMemset(pl->d_lastcol[w] + std::max<int64_t>(1, pl->total_T), 0, 16));   // the error word
        }
        for (int w = 0; w < kWorkspaces; ++w) {
            HIP_TRY(eng, hipEventCreate(&pl->ev_fill_done[w]));  // (kernel-attached events carry timestamps)
            HIP_TRY(eng, hipEventCreate(&pl->ev_bt_done[w]));
        }
        HIP_TRY(eng, hipStreamCreateWithFlags(&pl->side, hipStreamNonBlocking));
    }
    const int q = (int)(pl->pipe_runs % kWorkspaces);
    // Workspace q was last read by the backtrack of run k - kWorkspaces.  Back-pressure on the HOST: if that
    // backtrack is not finished, this call waits for it here (the caller is then a full rotation of runs ahead of
    // the GPU); the queue the fills run in never sees a wait between two of them.
    if (pl->bt_pending[q]) {
        if (hipEventQuery(pl->bt_done_ev[q]) != hipSuccess) HIP_TRY(eng, hipEventSynchronize(pl->bt_done_ev[q]));
        (void)hipGetLastError();   // (a "not ready" from the query is no error)
        pl->bt_pending[q] = false;
    }
    const bool timed = pl->ev_slots >= 2 * kWorkspaces && (pl->run_counter++ % pl->ev_stride == 0);
    hipEvent_t* ev = timed ? &pl->ev[(size_t)(pl->ev_runs % pl->ev_slots) * 4] : nullptr;
    // the events ride on the kernels' own dispatch packets: nothing else enters the queues
    hipEvent_t fill_done = ev ? ev[1] : pl->ev_fill_done[q];
    if ((rc = launch_fill(pl, a, q, st, ev ? ev[0] : nullptr, fill_done)) != CTCFA_OK) return rc;
    hipEvent_t bt_done = ev ? ev[3] : pl->ev_bt_done[q];
    if (pl->overlap_ok) {
        HIP_TRY(eng, hipStreamWaitEvent(pl->side, fill_done, 0));
        if ((rc = launch_backtrack(pl, a, want_seg, q, pl->side, ev ? ev[2] : nullptr, bt_done, true)) != CTCFA_OK) return rc;
    } else {   // (no room beside the next fill: behind this one, in the caller's stream)
        if ((rc = launch_backtrack(pl, a, want_seg, q, st, ev ? ev[2] : nullptr, bt_done, false)) != CTCFA_OK) return rc;
    }
    if (ev) pl->ev_runs++;
    pl->bt_done_ev[q] = bt_done;
    pl->bt_pending[q] = true;
    pl->pipe_runs++;
    return CTCFA_OK;
}

int ctcfa_plan_flush(ctcfa_plan* pl, void* stream) {
    if (!pl) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    GUARD_DEVICE(eng);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);  // NULL = the default (null) stream
    // `st` (any stream: the caller's, a copy or a collective stream) waits for every backtrack still in flight.
    // The workspaces stay marked as in use: the wait orders `st` only, not the stream the next fills are enqueued on,
    // so the host-side check of ctcfa_plan_run_pipelined (event query / synchronise) must still see them.
    for (int q = 0; q < kWorkspaces; ++q)
        if (pl->bt_pending[q]) {
            if (hipEventQuery(pl->bt_done_ev[q]) == hipSuccess) pl->bt_pending[q] = false;   // finished: nothing to order
            else HIP_TRY(eng, hipStreamWaitEvent(st, pl->bt_done_ev[q], 0));
        }
    (void)hipGetLastError();   // (a "not ready" from a query is no error)
    return CTCFA_OK;
}

int ctcfa_plan_get_timings(ctcfa_plan* pl, int n, float* fill_ms, float* backtrack_ms) {
    if (!pl || n <= 0 || !pl->ev_slots || n > pl->ev_slots || n > pl->ev_runs) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    GUARD_DEVICE(eng);
    for (int i = 0; i < n; ++i) {
        const int64_t run = pl->ev_runs - n + i;
        hipEvent_t* ev = &pl->ev[(size_t)(run % pl->ev_slots) * 4];
        HIP_TRY(eng, hipEventSynchronize(ev[3]));
        float a = 0.f, b = 0.f;
        HIP_TRY(eng, hipEventElapsedTime(&a, ev[0], ev[1]));
        HIP_TRY(eng, hipEventElapsedTime(&b, ev[2], ev[3]));
        if (fill_ms) fill_ms[i] = a;
        if (backtrack_ms) backtrack_ms[i] = b;
    }
    return CTCFA_OK;
}

int ctcfa_plan_get_step_intervals(ctcfa_plan* pl, int n, float* interval_ms) {
    // time from the start of the fill of one RECORDED run to the start of the fill of the next recorded run (n - 1
    // values for the last n recorded runs): with the timing stride s that is s steps of a back-to-back schedule
    if (!pl || !interval_ms || n < 2 || !pl->ev_slots || n > pl->ev_slots || n > pl->ev_runs) return CTCFA_ERR_INVALID;
    ctcfa_engine* eng = pl->eng;
    GUARD_DEVICE(eng);
    for (int i = 0; i + 1 < n; ++i) {
        const int64_t run = pl->ev_runs - n + i;
        hipEvent_t* e0 = &pl->ev[(size_t)(run % pl->ev_slots) * 4];
        hipEvent_t* e1 = &pl->ev[(size_t)((run + 1) % pl->ev_slots) * 4];
        HIP_TRY(eng, hipEventSynchronize(e1[0]));
        HIP_TRY(eng, hipEventElapsedTime(&interval_ms[i], e0[0], e1[0]));
    }
    return CTCFA_OK;
}

namespace {

hipError_t pinned_get(unsigned char** p, size_t* cap, size_t bytes) {
    if (*cap >= bytes) return hipSuccess;
    if (*p) (void)hipHostFree(*p);
    *p = nullptr;
    *cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(p), want, hipHostMallocDefault);
    if (e == hipSuccess) *cap = want;
    return e;
}

// Wide vocabularies (V > 256) through the staged kernels: per emission block, the columns its segments can look
// at -- the blank and the distinct labels of the block's texts -- renumbered 0 (blank), 1, 2, ... in ascending
// order of the original ids.  Possible when no block needs more than 256 of them (a 60 s window of a 5 000-piece
// model may; an anchor window of 30-60 tokens does not); the caller then runs the whole call on a compact
// [T, Vc] matrix (ctcfa::compact_kernel) and renumbered labels, and maps `state` back.
struct VocabRemap {
    int Vc = 0;                          // columns of the compact matrix
    std::vector<int32_t> orig;           // [emission blocks][Vc] original column of every compact column
    std::vector<int32_t> labels;         // renumbered label sequences, laid out like the caller's
    std::vector<int32_t> block_of;       // [batch] emission block of every segment
    std::vector<ctcfa::CompactBlock> blocks;
};

bool build_vocab_remap(int32_t batch, int32_t vocab, int32_t blank, const int32_t* T, const int32_t* C,
                       const int32_t* emission_of, const int32_t* labels, int max_cols, VocabRemap* out) {
    std::vector<int32_t> block_index(batch, -1);
    int nblocks = 0;
    out->block_of.assign(batch, 0);
    for (int b = 0; b < batch; ++b) {
        const int e = emission_of ? emission_of[b] : b;
        if (e == b) block_index[b] = nblocks++;
        out->block_of[b] = block_index[e];
    }
    std::vector<std::vector<int32_t>> sets(nblocks);
    int64_t lab_off = 0;
    for (int b = 0; b < batch; ++b) {
        std::vector<int32_t>& st = sets[out->block_of[b]];
        for (int c = 1; c < C[b]; ++c) {
            const int32_t g = labels[lab_off + c];
            if (g < 0 || g >= vocab) return false;   // (refused elsewhere; never index with it)
            if (g != blank) st.push_back(g);
        }
        lab_off += C[b];
    }
    int vc = 1;
    for (auto& st : sets) {
        std::sort(st.begin(), st.end());
        st.erase(std::unique(st.begin(), st.end()), st.end());
        vc = std::max(vc, (int)st.size() + 1);
    }
    if (vc > max_cols) return false;
    vc = std::max(vc, 2);
    out->Vc = vc;
    out->orig.assign((size_t)nblocks * vc, blank);
    for (int k = 0; k < nblocks; ++k)
        for (size_t i = 0; i < sets[k].size(); ++i) out->orig[(size_t)k * vc + 1 + i] = sets[k][i];
    out->labels.resize((size_t)lab_off);
    lab_off = 0;
    for (int b = 0; b < batch; ++b) {
        const std::vector<int32_t>& st = sets[out->block_of[b]];
        out->labels[lab_off] = labels[lab_off];   // the leading -1
        for (int c = 1; c < C[b]; ++c) {
            const int32_t g = labels[lab_off + c];
            out->labels[lab_off + c] = g == blank ? 0 : 1 + (int32_t)(std::lower_bound(st.begin(), st.end(), g) - st.begin());
        }
        lab_off += C[b];
    }
    out->blocks.resize(nblocks);
    int64_t src = 0, dst = 0;
    for (int b = 0; b < batch; ++b)
        if (block_index[b] >= 0) {
            out->blocks[block_index[b]] = ctcfa::CompactBlock{src, dst, T[b], 0};
            src += (int64_t)T[b] * vocab;
            dst += (int64_t)T[b] * vc;
        }
    return true;
}

// One synchronous alignment call on the engine's stream (or the caller's): transient plan, ONE packed
// upload of the small inputs, the emissions uploaded (host_lpz) or used where they are (dev_lpz), both
// kernels, ONE result download.  An anchor iteration issues hundreds of such calls for windows of a
// few hundred frames: what a call costs beyond its kernels is what this function keeps small.
int align_impl(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab, const int32_t* T,
               const int32_t* C, const int32_t* U, const int32_t* emission_of, int32_t label_width,
               const float* host_lpz, const float* dev_lpz, hipStream_t st,
               const int32_t* labels, const int32_t* utt_begin, int32_t* frame_of_label, float* char_prob,
               int32_t* state, double* seg_start, double* seg_end, double* seg_score, int32_t* t_end,
               int32_t* status) {
    GUARD_DEVICE(eng);
    // CTCFA_CALL_TRACE=1 (tuning): where the host time of a call goes, printed when the engine is destroyed
    static const bool trace = std::getenv("CTCFA_CALL_TRACE") != nullptr;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto tp = now();
    auto lap = [&](int k) {
        if (!trace) return;
        const auto t = now();
        eng->trace_ns[k] += std::chrono::duration_cast<std::chrono::nanoseconds>(t - tp).count();
        tp = t;
    };
    // vocabularies above 128 entries: the compact matrix of the columns each emission block can look at, if that fits
    VocabRemap remap;
    const int32_t wide_vocab = vocab;
    ctcfa_params compact_params;
    bool compact = false;
    // Worth it where the gather kernel cannot serve (its flags: the package's defaults only; 961 label columns) or is
    // slow (long texts: 1.9 ms against 0.4 for config 3's shape at 128 columns); short anchor windows are as fast
    // through the gather kernel (tools/wide_vocab_timing.py) and skip the pre-pass.  CTCFA_REMAP=1 forces it.
    bool remap_wanted = false;
    const bool gratis_call = params && (params->flags & CTCFA_FLAG_BLANK_TRANSITION_COST_ZERO);
    // (blank_transition_cost_zero over 65 .. 256 entries: a narrowed plan takes the emissions as they are -- tried first)
    ctcfa_plan* pl = nullptr;
    int rc = CTCFA_ERR_UNSUPPORTED;
    if (vocab > 64 && vocab <= kMaxStagedVocab && gratis_call && label_width == 1 && labels && T && C &&
        !std::getenv("CTCFA_NO_NARROW") && !std::getenv("CTCFA_REMAP"))
        rc = plan_create_impl(eng, &pl, params, batch, vocab, T, C, U, 0, true, emission_of, labels, label_width);
    if (rc == CTCFA_OK) {
    } else if (vocab > kMaxStagedVocab && label_width == 1 && params && T && C) {
        int cmax = 0;
        for (int b = 0; b < batch; ++b) cmax = std::max(cmax, (int)C[b]);
        remap_wanted = cmax >= 192 || !(params->flags & CTCFA_FLAG_PREAMBLE_TRANSITION_COST_ZERO) || gratis_call ||
                       std::getenv("CTCFA_REMAP");
    } else if (vocab > 64 && gratis_call && label_width == 1 && T && C) {
        remap_wanted = true;   // (checkpoint mode, the only one that takes the flag, stages at most 64 columns)
    }
    if (remap_wanted && !std::getenv("CTCFA_NO_REMAP") && !std::getenv("CTCFA_FORCE_GATHER") &&
        params->blank >= 0 && params->blank < vocab &&
        build_vocab_remap(batch, vocab, params->blank, T, C, emission_of, labels, gratis_call ? 64 : kMaxStagedVocab, &remap)) {
        compact = true;
        compact_params = *params;
        compact_params.blank = 0;
        params = &compact_params;
        labels = remap.labels.data();
        vocab = remap.Vc;
    }
    if (rc != CTCFA_OK)
        rc = plan_create_impl(eng, &pl, params, batch, vocab, T, C, U, 0, true, emission_of,
                              label_width > 1 ? nullptr : labels, label_width);
    if (rc != CTCFA_OK) return rc;
    lap(0);
    const bool want_seg = U && utt_begin && seg_start && seg_end && seg_score && pl->total_U > 0;
    const size_t n_lpz = (size_t)pl->total_lpz_T * wide_vocab, n_lab = (size_t)pl->total_C, n_frm = (size_t)pl->total_T;
    const size_t n_utt = (size_t)pl->total_U, n_ub = n_utt + batch;
    auto cleanup = [&]() {  // the buffers are the engine's scratch; quiesce before the next call reuses them
        (void)hipStreamSynchronize(st);
        ctcfa_plan_destroy(pl);
    };
#define AB_TRY(expr)                                                                      \
    do {                                                                                  \
        hipError_t _e = (expr);                                                           \
        if (_e != hipSuccess) {                                                           \
            std::string m = std::string(#expr) + ": " + hipGetErrorString(_e);            \
            cleanup();                                                                    \
            return set_err(eng, CTCFA_ERR_HIP, m);                                        \
        }                                                                                 \
    } while (0)
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    // packed upload: role table | segment table | labels | utterance starts | watch columns of shared fills
    const size_t n_watch = pl->watch.size();
    const size_t in_roles = 0, in_segs = up(sizeof(ctcfa::FillRoles)), in_lab = in_segs + up(sizeof(SegDesc) * (size_t)batch),
                 in_ub = in_lab + up(n_lab * 4 * (size_t)pl->S), in_watch = in_ub + (want_seg ? up(n_ub * 4) : 0),
                 in_orig = in_watch + up(n_watch * sizeof(ctcfa::WatchDesc)),
                 in_cblk = in_orig + (compact ? up(remap.orig.size() * 4) : 0),
                 in_bytes = in_cblk + (compact ? up(remap.blocks.size() * sizeof(ctcfa::CompactBlock)) : 0);
    const size_t o_fol = 0, o_cp = o_fol + up(n_lab * 4), o_state = o_cp + up(n_frm * 4),
                 o_tend = o_state + (state ? up(n_frm * 4) : 0), o_status = o_tend + up((size_t)batch * 4),
                 o_seg = o_status + up((size_t)batch * 4), out_bytes = o_seg + (want_seg ? up(3 * n_utt * 8) : 0);
    float* d_lpz = const_cast<float*>(dev_lpz);
    unsigned char *d_in = nullptr, *d_out = nullptr;
    if (!d_lpz) AB_TRY(scratch_get(eng, kSlotLpz, reinterpret_cast<void**>(&d_lpz), n_lpz * sizeof(float)));
    AB_TRY(scratch_get(eng, kSlotIn, reinterpret_cast<void**>(&d_in), in_bytes));
    AB_TRY(pinned_get(&eng->h_in, &eng->h_in_cap, in_bytes));
    AB_TRY(pinned_get(&eng->h_out, &eng->h_out_cap, out_bytes));
    // Small calls (a window of the anchor iteration: a few KB of results): the backtrack writes its results straight into
    // the pinned result block -- posted writes over PCIe -- and the download, a copy packet of its own in the queue
    // (~7 us between the last kernel and the host), goes away.  Only where the last kernel never reads its outputs back
    // (the checkpoint-mode backtrack keeps its own copies in LDS; the decision-word one scores from frame_of_label in
    // memory, the windowed and rescoring kernels read what the backtrack wrote).
    static const bool no_direct = std::getenv("CTCFA_NO_DIRECT_OUT") != nullptr;
    const bool direct_out = !no_direct && out_bytes <= (size_t)(96 << 10) && pl->ckpt && pl->win_list.empty() &&
                            pl->prm.score_min_mean_over_L <= 128;
    if (direct_out) AB_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_out), eng->h_out, 0));
    else AB_TRY(scratch_get(eng, kSlotOut, reinterpret_cast<void**>(&d_out), out_bytes));
    unsigned char* h = eng->h_in;
    std::memcpy(h + in_roles, &pl->roles, sizeof(ctcfa::FillRoles));
    std::memcpy(h + in_segs, pl->segs.data(), sizeof(SegDesc) * (size_t)batch);
    std::memcpy(h + in_lab, labels, n_lab * 4 * (size_t)pl->S);
    if (want_seg) std::memcpy(h + in_ub, utt_begin, n_ub * 4);
    if (n_watch) std::memcpy(h + in_watch, pl->watch.data(), n_watch * sizeof(ctcfa::WatchDesc));
    if (compact) {
        std::memcpy(h + in_orig, remap.orig.data(), remap.orig.size() * 4);
        std::memcpy(h + in_cblk, remap.blocks.data(), remap.blocks.size() * sizeof(ctcfa::CompactBlock));
    }
    lap(1);
    AB_TRY(hipMemcpyAsync(d_in, h, in_bytes, hipMemcpyHostToDevice, st));
    if (host_lpz) AB_TRY(hipMemcpyAsync(d_lpz, host_lpz, n_lpz * sizeof(float), hipMemcpyHostToDevice, st));
    if (compact) {   // the staged kernels run on the compact matrix from here on
        float* d_compact = nullptr;
        AB_TRY(scratch_get(eng, kSlotCompact, reinterpret_cast<void**>(&d_compact), (size_t)pl->total_lpz_T * remap.Vc * sizeof(float)));
        int tmax = 1;
        for (const auto& cb : remap.blocks) tmax = std::max(tmax, (int)cb.T);
        const unsigned chunks = (unsigned)std::max<int64_t>(1, std::min<int64_t>(64, ((int64_t)tmax * remap.Vc + 4095) / 4096));
        hipLaunchKernelGGL(ctcfa::compact_kernel, dim3((unsigned)remap.blocks.size(), chunks), dim3(256), 0, st,
                           reinterpret_cast<const ctcfa::CompactBlock*>(d_in + in_cblk), (const float*)d_lpz,
                           reinterpret_cast<const int32_t*>(d_in + in_orig), (int)wide_vocab, remap.Vc, d_compact);
        AB_TRY(hipGetLastError());
        d_lpz = d_compact;
    }
    lap(2);
    pl->d_roles = reinterpret_cast<ctcfa::FillRoles*>(d_in + in_roles);
    pl->d_segs = reinterpret_cast<SegDesc*>(d_in + in_segs);
    pl->d_watch = n_watch ? reinterpret_cast<ctcfa::WatchDesc*>(d_in + in_watch) : nullptr;
    int32_t* d_lab = reinterpret_cast<int32_t*>(d_in + in_lab);
    int32_t* d_ub = want_seg ? reinterpret_cast<int32_t*>(d_in + in_ub) : nullptr;
    double* d_seg = want_seg ? reinterpret_cast<double*>(d_out + o_seg) : nullptr;
    // the backtrack is this call's last kernel and writes into the pinned block: it also says when it is done
    const bool done_word = direct_out && eng->d_done_word != nullptr;
    if (done_word) {
        eng->done_total += (uint32_t)pl->B;   // (one backtrack workgroup per segment)
        eng->done_seq += 1u;
        if (eng->done_seq == 0u) eng->done_seq = 1u;
        pl->done_target = eng->done_total;
        pl->done_value = eng->done_seq;
    }
    rc = ctcfa_plan_run_device(pl, d_lpz, d_lab, d_ub, reinterpret_cast<int32_t*>(d_out + o_fol),
                               reinterpret_cast<float*>(d_out + o_cp),
                               state ? reinterpret_cast<int32_t*>(d_out + o_state) : nullptr, d_seg,
                               d_seg ? d_seg + n_utt : nullptr, d_seg ? d_seg + 2 * n_utt : nullptr,
                               reinterpret_cast<int32_t*>(d_out + o_tend), reinterpret_cast<int32_t*>(d_out + o_status), st);
    if (rc != CTCFA_OK) {
        if (done_word) eng->d_done_word = nullptr;   // (the device counter may not have moved: no more completion words on this engine)
        cleanup();
        return rc;
    }
    lap(3);
    unsigned char* ho = eng->h_out;
    if (!direct_out) AB_TRY(hipMemcpyAsync(ho, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    lap(4);
    bool waited = false;
    if (done_word) {   // poll the completion word (the stream's own signal arrives ~3 us later); a second without it: the stream
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spins = 1;; ++spins) {
            if (__atomic_load_n(eng->h_done, __ATOMIC_ACQUIRE) == pl->done_value) {
                waited = true;
                break;
            }
            if ((spins & 1023u) == 0u && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(1)) break;
        }
    }
    if (!waited) {
        if (done_word) eng->d_done_word = nullptr;   // (a word that did not come: this engine waits for its streams from now on)
        AB_TRY(hipStreamSynchronize(st));
    }
    lap(5);
    std::memcpy(frame_of_label, ho + o_fol, n_lab * 4);
    std::memcpy(char_prob, ho + o_cp, n_frm * 4);
    if (state) {
        std::memcpy(state, ho + o_state, n_frm * 4);
        if (compact) {   // label ids of the caller's vocabulary again (-1 / -2: self transition / untouched)
            int64_t f = 0;
            for (int b = 0; b < batch; ++b) {
                const int32_t* cols = remap.orig.data() + (size_t)remap.block_of[b] * remap.Vc;
                for (int t = 0; t < T[b]; ++t, ++f)
                    if (state[f] >= 0) state[f] = cols[state[f]];
            }
        }
    }
    std::memcpy(t_end, ho + o_tend, (size_t)batch * 4);
    std::memcpy(status, ho + o_status, (size_t)batch * 4);
    if (want_seg) {
        std::memcpy(seg_start, ho + o_seg, n_utt * 8);
        std::memcpy(seg_end, ho + o_seg + n_utt * 8, n_utt * 8);
        std::memcpy(seg_score, ho + o_seg + 2 * n_utt * 8, n_utt * 8);
    }
#undef AB_TRY
    ctcfa_plan_destroy(pl);
    lap(6);
    if (trace) ++eng->trace_calls;
    return CTCFA_OK;
}

}  // namespace

int ctcfa_align_batch(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                      const int32_t* T, const int32_t* C, const int32_t* U, const float* lpz,
                      const int32_t* labels, const int32_t* utt_begin, int32_t* frame_of_label,
                      float* char_prob, int32_t* state, double* seg_start, double* seg_end,
                      double* seg_score, int32_t* t_end, int32_t* status) {
    if (!eng) return set_err(nullptr, CTCFA_ERR_INVALID, "engine == NULL");
    if (!lpz || !labels || !frame_of_label || !char_prob || !t_end || !status)
        return set_err(eng, CTCFA_ERR_INVALID, "NULL host buffer");
    return align_impl(eng, params, batch, vocab, T, C, U, nullptr, 1, lpz, nullptr, eng->stream, labels, utt_begin,
                      frame_of_label, char_prob, state, seg_start, seg_end, seg_score, t_end, status);
}

int ctcfa_align_batch_resident(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                               const int32_t* T, const int32_t* C, const int32_t* U, const float* d_lpz,
                               const int32_t* labels, const int32_t* utt_begin, int32_t* frame_of_label,
                               float* char_prob, int32_t* state, double* seg_start, double* seg_end,
                               double* seg_score, int32_t* t_end, int32_t* status, void* stream) {
    if (!eng) return set_err(nullptr, CTCFA_ERR_INVALID, "engine == NULL");
    if (!d_lpz || !labels || !frame_of_label || !char_prob || !t_end || !status)
        return set_err(eng, CTCFA_ERR_INVALID, "NULL buffer");
    return align_impl(eng, params, batch, vocab, T, C, U, nullptr, 1, nullptr, d_lpz, reinterpret_cast<hipStream_t>(stream),
                      labels, utt_begin, frame_of_label, char_prob, state, seg_start, seg_end, seg_score, t_end, status);
}

int ctcfa_align_batch_shared(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                             const int32_t* T, const int32_t* C, const int32_t* U, const int32_t* emission_of,
                             const float* lpz, int32_t lpz_on_device, const int32_t* labels,
                             const int32_t* utt_begin, int32_t* frame_of_label, float* char_prob, int32_t* state,
                             double* seg_start, double* seg_end, double* seg_score, int32_t* t_end,
                             int32_t* status, void* stream) {
    if (!eng) return set_err(nullptr, CTCFA_ERR_INVALID, "engine == NULL");
    if (!lpz || !labels || !frame_of_label || !char_prob || !t_end || !status || !emission_of)
        return set_err(eng, CTCFA_ERR_INVALID, "NULL buffer");
    return align_impl(eng, params, batch, vocab, T, C, U, emission_of, 1, lpz_on_device ? nullptr : lpz,
                      lpz_on_device ? lpz : nullptr, lpz_on_device ? reinterpret_cast<hipStream_t>(stream) : eng->stream,
                      labels, utt_begin, frame_of_label, char_prob, state, seg_start, seg_end, seg_score, t_end, status);
}

int ctcfa_align_batch_spans(ctcfa_engine* eng, const ctcfa_params* params, int32_t batch, int32_t vocab,
                            int32_t label_width, const int32_t* T, const int32_t* C, const int32_t* U,
                            const float* lpz, const int32_t* label_matrix, const int32_t* utt_begin,
                            int32_t* frame_of_label, float* char_prob, int32_t* state, double* seg_start,
                            double* seg_end, double* seg_score, int32_t* t_end, int32_t* status) {
    if (!eng) return set_err(nullptr, CTCFA_ERR_INVALID, "engine == NULL");
    if (!lpz || !label_matrix || !frame_of_label || !char_prob || !t_end || !status)
        return set_err(eng, CTCFA_ERR_INVALID, "NULL host buffer");
    return align_impl(eng, params, batch, vocab, T, C, U, nullptr, label_width, lpz, nullptr, eng->stream, label_matrix,
                      utt_begin, frame_of_label, char_prob, state, seg_start, seg_end, seg_score, t_end, status);
}

int ctcfa_plan_create_shared(ctcfa_engine* eng, ctcfa_plan** out, const ctcfa_params* params, int32_t batch,
                             int32_t vocab, const int32_t* T, const int32_t* C, const int32_t* U,
                             const int32_t* emission_of, const int32_t* labels, int32_t force_k) {
    return plan_create_impl(eng, out, params, batch, vocab, T, C, U, force_k, false, emission_of, labels);
}

int ctcfa_plan_get_sharing(const ctcfa_plan* pl, int32_t* n_fills, int32_t* n_emission_blocks) {
    if (!pl) return CTCFA_ERR_INVALID;
    if (n_fills) {
        int n = 0;
        for (const SegDesc& s : pl->segs) n += (s.prestatus == CTCFA_ST_OK && !s.fill_skip);
        *n_fills = n;
    }
    if (n_emission_blocks) *n_emission_blocks = pl->n_emission_blocks;
    return CTCFA_OK;
}

}  // extern "C"
