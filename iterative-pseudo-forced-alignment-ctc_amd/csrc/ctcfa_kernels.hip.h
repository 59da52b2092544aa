// ctcfa_kernels.hip.h -- gfx950 device code of the CTC forced-alignment engine.
//
// What is computed (per segment; single labels, S = 1, unless said otherwise):
//   table[t,c] = max( table[t-1,c-1] + lpz[t,g_c],  table[t-1,c] + max(lpz[t,blank], lpz[t,g_c]),  -1e9 )
// in fp32, exactly the recurrence of ctc-segmentation 1.7.1's cython_fill_table
// (requirements.txt:13; reached from src/iterative_utterance_alignment.py:216), plus the
// transition the package's backtrack would infer at (t,c) from fp32 residuals:
//   SWITCH  iff  | max(lb,e) - (table[t,c]-table[t-1,c]) |  >  | e - (table[t,c]-table[t-1,c-1]) |
// Decision-word mode: that predicate is evaluated at fill time, while all four operands are in
// registers, and kept as ONE BIT per cell.  Checkpoint mode (vocabulary <= 64, chosen per plan):
// the fill keeps only the table row every 32-row block ends in (the same 32 bits per block and
// column) and the backtrack re-runs recurrence + predicate over the 64 columns x 32 rows around
// the path.  Either way the fp32 trellis itself never leaves the chip.
//
// Kernels:
//   fill_kernel<K,VP,CK>      T <= min_window_size, vocabulary <= 128: the hot one
//   fill_gather_kernel        same, vocabulary > 128 (no LDS staging of vocabulary rows)
//   backtrack_kernel          decision-word mode: end cell, walk over the decision words, per-frame outputs, scores
//   stride_backtrack_kernel   checkpoint mode: the same outputs, the decisions recomputed from the checkpoint rows by
//                             waves that work on 32-row blocks speculatively ("striders")
//   windowed_kernel           T > min_window_size: the package's windowed regime, literally; also
//                             every segment with a label matrix (multi-character tokens, S > 1)
//
// Mapping of fill_kernel (CDNA4, 64-wide waves, no MFMA: a max-plus scan is not a contraction):
//   * one workgroup per segment; its padded columns are cut into tiles, a compute wave owns one
//     tile, lane l owns K consecutive columns -> the c-1 neighbour is in-lane for k>0 and one DPP
//     wave_shr:1 (folded into the add) for k==0.  The first HL lanes of a tile are its HALO: copies
//     of the left neighbour's last HL*K >= 16 columns, which go wrong one column per row and are
//     refreshed every 16 rows (a "group") from an LDS exchange ring -- no per-row traffic between
//     tiles and no pipeline skew.  A role table (FillRoles) says what each wave is: tile, producer
//     (1 or 2), idle padding;
//   * no workgroup barrier after start-up: waves pace each other through counters in LDS (done[w],
//     staged[p]); every wait is bounded (a lost counter becomes status 5, never a hang);
//   * emission rows are staged once per workgroup through an LDS ring of 32-row blocks as (e, m)
//     pairs, m = max(lb, e, -1e9), row pitch VP+2 entries; a gather is one ds_read_b64; entry VP is
//     the "start column" pseudo-label (e = -inf, m = 0) that makes column 0 (ground_truth == -1) and
//     the left padding reproduce table[t,0]; blank_transition_cost_zero = the blank entry's m is 0;
//   * decisions are shifted into a per-(lane,k) register (v_alignbit) and stored every 32 rows as
//     words bits[block][column] (checkpoint mode: the table value of the block's last row goes there
//     instead); HBM traffic per segment is 4*T*V (emissions, once) + T*Cpad/8 (trace words) + 4*T
//     (last-column scores);
//   * the tile that holds the last label column publishes its score every row (for the end cell's
//     argmax); a SHARED fill (texts that are prefixes of one another over the same emissions) does
//     the same for every member's last column ("watch columns");
//   * blocks that cannot matter are skipped exactly (dead zone behind the end cell's cone -- per
//     tile against the nearest watch column; the -1e9 zone above the diagonal while every emission
//     so far is <= 0).
#pragma once
#include <hip/hip_runtime.h>
#ifndef CTCFA_PRODUCER_PRIO
#define CTCFA_PRODUCER_PRIO 1   // above the backtrack waves of the previous batch (priority 0), below the tiles
#endif
#ifndef CTCFA_TILE_PRIO_BASE
#define CTCFA_TILE_PRIO_BASE 2   // earlier half of a segment's tiles; the later half runs one above
#endif
#ifndef CTCFA_VGPR_CAP
#define CTCFA_VGPR_CAP 1
#endif
#ifndef CTCFA_TWO_PROD32
#define CTCFA_TWO_PROD32 0   // tuning: two producer waves for the 32-entry vocabulary too (each stages every other group of 8 rows)
#endif
#ifndef CTCFA_ABL
#define CTCFA_ABL 0   // tuning builds only (results are WRONG for != 0), a bit mask of what is left out, to price it: 1 the group hand-over
                      // (exchange rows, counters, waits for the neighbour), 2 the waits between tiles and producer (staged / ring space),
                      // 4 the trace-word and last-column stores, 8 the producer's work (it leaves at once; needs 2)
#endif
#ifndef CTCFA_ADDTID_PRODUCER
#define CTCFA_ADDTID_PRODUCER 1   // the 32-entry vocabulary: two producers, one ds_write_addtid_b32 per ring row (0: one producer, 16-byte stores)
#endif
#ifndef CTCFA_PROD_PACE
#define CTCFA_PROD_PACE 0   // s_sleep argument (x 64 clocks) between two LDS stores of a producer in the steady state (0: back to back)
#endif
// Round 4's changes to the tile loop, each behind a switch (measured on one box, profiles/r04_fill_variants.txt: fill alone |
// pipelined step, config 3: round 3 142.7 us | 0.1605 ms; lean hand-over alone 139.7 | 0.1592; + two-block bodies 135.6 | 0.1604;
// + every lane publishing 135.7 | 0.1646; + the owner's stores deferred 137.3 | 0.1637 -- and for the 38-entry vocabulary every
// one of them LOSES against round 3's loop, 171..175 us against 164): shipped = lean hand-over and two-block bodies for the
// 32-entry pitch, round 3's loop for every other pitch.
#ifndef CTCFA_LEAN_HANDOVER
#define CTCFA_LEAN_HANDOVER 1    // 32-entry pitch: counter and exchange row read together 4 rows before a group's end, looked at at its end; 0 = rounds 2-3 everywhere
#endif
#ifndef CTCFA_LEAN_ALL_PITCHES
#define CTCFA_LEAN_ALL_PITCHES 0  // tuning: the lean hand-over and the two-block bodies for every (e, m)-pair pitch up to 64 entries
#endif
#ifndef CTCFA_MASKED_PUBLISH
#define CTCFA_MASKED_PUBLISH 1   // 1 = two-column tiles store their exchange row under an exec mask (rounds 2-3); 0 = every lane stores (the rest into the sink)
#endif
#ifndef CTCFA_OWNER_DEFER
#define CTCFA_OWNER_DEFER 0      // 1 = the owner tile stores a row's score one row later and copies a block's scores out in the middle of the next block
#endif
#ifndef CTCFA_BODY_BLOCKS
#define CTCFA_BODY_BLOCKS 2   // 32-row blocks per body of the tile loop where the ring allows it (1: one block per body, as up to round 3)
#endif
#ifndef CTCFA_PF
#define CTCFA_PF 2  // rows of LDS prefetch distance in the fill kernel
#ifndef CTCFA_STAMP_BLOCK
#define CTCFA_STAMP_BLOCK 40  // (CTCFA_STAMP=3: the block, and the one after it, that get a stamp every 8 rows)
#endif
#endif
// Trace words are written once and read by another kernel (another XCD, as likely as not): stored past the L2
// (nt), they leave no dirty lines for the end-of-kernel write-back to find between two fills.
#ifndef CTCFA_TRACE_NT
#define CTCFA_TRACE_NT 1
#endif
#if CTCFA_TRACE_NT
#define CTCFA_TRACE_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define CTCFA_TRACE_STORE(ptr, val) (*(ptr) = (val))
#endif
#include <stdint.h>
#include <type_traits>
#include <utility>

namespace ctcfa {

constexpr int kRows = 32;      // rows per block == bits per decision word
constexpr int kBnd = 128;      // ring length (rows) of the cross-wave boundary column
constexpr int kBndPitch = kBnd; // floats per boundary ring: index t % 128 holds row t
constexpr int kSinkBytes = 1280; // sink for lanes that publish nothing (64 x 16 B, plus slack)
constexpr int kPitchPad = 2;   // LDS row = VP + 2 entries: 16-B aligned rows, banks rotate by 4 per row
constexpr float kProbMax = -1000000000.0f;   // Cython sentinel (prob_max)
constexpr float kMaxProb = -10000000000.0f;  // config.max_prob on the NumPy side
constexpr int kPreWindowed = 100;  // SegDesc.prestatus: T > min_window_size, handled by windowed_kernel

struct SegDesc {
    int64_t lpz_off;   // elements into lpz
    int64_t lab_off;   // elements into labels / frame_of_label
    int64_t frm_off;   // elements into char_prob / state / lastcol
    int64_t utt_off;   // elements into seg_*; utt_begin is at utt_off + b
    int64_t bits_off;  // words into the decision-bit workspace
    int32_t T, C, U;
    int32_t shift;      // left padding so that column C-1 lands on k == K-1
    int32_t prestatus;  // 0, or the status decided from shapes alone
    int32_t seg_index;
    int32_t owner_stage;  // pipeline stage (tile) that holds the last label column C-1
    int32_t owner_lane;   // ... and the lane inside it (the column sits at k == K-1 there)
    int64_t win_off;      // windowed segments: floats into the table workspace
    int64_t wcol_off;     // windowed segments: ints into the per-column offsets workspace
    // Shared fills (ctcfa_*_shared): segments over the same emissions whose label sequences are prefixes
    // of the group's longest one.  The longest (the LEADER) is filled; every member's last label column
    // is a WATCH column of that fill (its scores go to lastcol + the member's frm_off), and the members'
    // backtracks walk the leader's trace words (same bits_off, same shift).
    int32_t watch_first;  // leader: first entry of this group in the watch table
    int32_t watch_n;      // leader: entries (0: not a group -- only column C-1, by owner_stage / owner_lane)
    int32_t fill_skip;    // follower: no fill workgroup of its own
    int32_t reserved0;
};

struct WatchDesc {     // one watched label column of a shared fill, ascending pcol within a group
    int64_t frm_off;   // where its scores go: lastcol + frm_off + t
    int32_t pcol;      // padded column (label column + the leader's shift)
    int32_t reserved;
};
constexpr int kMaxWatch = 16;        // watch columns per shared fill (more members: fills of their own)
#ifndef CTCFA_WATCH_MAX_K
#define CTCFA_WATCH_MAX_K 2
#endif
constexpr int kWatchMaxK = CTCFA_WATCH_MAX_K;        // widest tile the watch variant of the row loop is compiled for (wider ones are at their VGPR cap)

// Launch shape of the fill kernel: what each wave of a workgroup does.  Tiles are numbered left
// to right; a tile is one wave, lane l owns K consecutive padded columns.  The first HL lanes of a
// tile are its HALO: copies of the last HL*K columns of the tile to its left.
enum : int8_t { kRoleIdle = 0, kRoleProducer = 1, kRoleTile = 2 };
struct WaveRole {
    int8_t role;     // low nibble: kRole*; high nibble: the wave's issue priority (s_setprio 0..3), set by the host's role table
    int8_t stage;    // tile index (tiles), or the share of the rows a producer stages
    int16_t cbase;   // padded column of lane 0, k = 0 (tile 0: negative, its halo is left padding)
};
__host__ __device__ constexpr int8_t role_with_prio(int role, int prio) { return (int8_t)(role | (prio << 4)); }
__device__ __forceinline__ void set_wave_prio(int p) {   // (s_setprio takes an immediate)
    if (p <= 0) __builtin_amdgcn_s_setprio(0);
    else if (p == 1) __builtin_amdgcn_s_setprio(1);
    else if (p == 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(3);
}
struct FillRoles {
    int32_t nwaves;   // waves per workgroup (blockDim.x / 64)
    int32_t nstages;  // compute tiles W
    int32_t cpad;     // padded columns = W * (64 - HL) * K (row pitch of the trace words)
    int32_t nslots;   // emission ring slots NS (32-row blocks)
    int32_t nprod;    // producer waves (1, or 2: each stages every other row)
    int32_t reserved[3];
    WaveRole wave[16];
};

constexpr int kHaloRows = 16;       // rows between two refreshes of a tile's halo (a "group")
constexpr int kGroups = kRows / kHaloRows;
#ifndef CTCFA_POLL_LEAD
#define CTCFA_POLL_LEAD 4
#endif
#ifndef CTCFA_NBR_SLEEP
#define CTCFA_NBR_SLEEP 1
#endif
constexpr int kPollLead = CTCFA_POLL_LEAD;   // a tile reads its neighbour's counter and exchange row this many rows before the group's end (where it looks at them)
constexpr int kExchangeRing = 8;    // group slots of a tile's exchange ring (ring slots NS <= 4, two groups per block)
constexpr int kFlagInts = 32;       // done[16], staged[2], posflag, always-done [19], a bounded wait gave up [20], pad
constexpr int kBigCount = 0x3fffffff;
constexpr int kSpinCap = 1 << 20;     // every wait gives up after ~0.1 s: a lost counter must not hang the GPU
#ifdef CTCFA_DEBUG_SPIN
#define CTCFA_SPIN_DIAG(what, a, b, c) do { if ((threadIdx.x & 63) == 0) printf("fill spin timeout: %s wg %d a %d b %d c %d T %d C %d\n", what, (int)blockIdx.x, (int)(a), (int)(b), (int)(c), T, C); } while (0)
#else
// a wait on a progress counter gave up (a lost counter would otherwise hang the GPU): flags[20] is raised HERE (an LDS word: no vector-memory operation in the loops the
// compiler would then drain vmcnt for -- with one in a producer's bounded wait it waited for every load in flight at the
// next join, taken or not), and every wave that leaves looks at it and writes this run's number into the workspace's
// error word (CTCFA_SPIN_REPORT); the backtrack of the same run turns that into status CTCFA_ST_INTERNAL for the batch
#define CTCFA_SPIN_DIAG(what, a, b, c) do { if ((threadIdx.x & 63) == 0) flags[20] = 1; } while (0)
#endif

using lds_vint = volatile __attribute__((address_space(3))) int;   // counters in LDS

// ---------------------------------------------------------------------------------------
// NARROWED plans.  The vocabulary has 33 .. 256 entries, but no segment's text uses more than 31 of them beside the blank
// (a character model's window: the reference's 38-token model).  Fill and checkpoint-mode backtrack then stage 32 entries of
// an emission row -- the RING entries of the segment: entry 0 the blank, entry r >= 1 the r-th smallest vocabulary entry its
// text uses -- and run at the 32-entry pace whatever the vocabulary.  Every wave derives the table from the labels by
// itself (48 words of LDS that nobody else touches meanwhile, and 256 bytes behind them while it is built; LDS executes a
// wave's operations in order: no barrier):
//   [0..7]   the set of vocabulary entries the text uses (the blank left out), a bit each
//   [8..15]  how many of them lie below each of the eight words
//   [16..47] ring entry -> vocabulary entry (entries the text leaves free: the blank)
// A text with more labels than that breaks the caller's promise (CTCFA_FLAG_TEXTS_OF_31_LABELS): the backtrack reports
// CTCFA_ST_TOO_MANY_LABELS for the segment; the fill keeps every address inside the ring (`& 31`) and its results are dropped.
// ---------------------------------------------------------------------------------------
// A ring of 64 entries (entry 63 stays free: the 64-entry producers keep lane 63 for the start column's pseudo entry) takes
// texts of up to 62 labels over vocabularies of 65 .. 256 entries through the 64-entry kernels the same way: the table then has
// 64 words ([16..79]).
constexpr int kNarrowWords = 16 + 64;
constexpr int kNarrowScratchBytes = kNarrowWords * 4 + 256;   // ... and, while the table is built, a byte per vocabulary entry behind it
__host__ __device__ constexpr int narrow_max_labels(int ring) { return ring == 32 ? 31 : 62; }   // labels beside the blank a ring takes
using lds_vuint = volatile __attribute__((address_space(3))) uint32_t;
using lds_vbyte = volatile __attribute__((address_space(3))) uint8_t;
template <int RING = 32>
__device__ __forceinline__ void narrow_build(lds_vuint* scr, const int32_t* __restrict__ seg_lab, int C, int V, int blank, int lane) {
    // which entries the text uses: a byte each, set by plain stores (lanes that write the same byte write the same value;
    // LDS atomics on eight words serialise the 64 lanes of every instruction: ~9 us for config 3's workgroups, all at once)
    lds_vbyte* used = (lds_vbyte*)(scr + kNarrowWords);
    scr[kNarrowWords + lane] = 0u;
    if (lane < RING) scr[16 + lane] = static_cast<uint32_t>(blank);
    // (sixteen loads in flight: one memory round trip for a text of up to 1 024 labels)
    for (int c0 = 1; c0 < C; c0 += 64 * 16) {
        int g[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int c = c0 + lane + 64 * u;
            g[u] = seg_lab[c < C ? c : C - 1];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (g[u] != blank) used[g[u] & 255] = 1;
    }
    uint32_t word = 0u;   // lanes 0 .. 7: word `lane` of the set
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint64_t m = __builtin_amdgcn_ballot_w64(used[lane + 64 * i] != 0);
        if ((lane >> 1) == i) word = (lane & 1) ? static_cast<uint32_t>(m >> 32) : static_cast<uint32_t>(m);
    }
    const int cnt = __builtin_popcount(word);
    int below = 0;
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int ci = __builtin_amdgcn_readlane(cnt, i);
        below += lane > i ? ci : 0;
    }
    if (lane < 8) {
        scr[lane] = word;
        scr[8 + lane] = static_cast<uint32_t>(below);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = lane + 64 * i;
        const uint32_t wd = scr[g >> 5];
        if ((wd >> (g & 31)) & 1u) {
            const int r = 1 + static_cast<int>(scr[8 + (g >> 5)]) + __builtin_popcount(wd & ((1u << (g & 31)) - 1u));
            if (r < RING) scr[16 + r] = static_cast<uint32_t>(g < V ? g : V - 1);
        }
    }
    // the entries the text leaves free: the vocabulary entries behind its last one, as far as there are any (nobody reads what
    // is staged there; neighbours in the vocabulary row let the backtrack's staging load four ring entries at once)
    const int used_n = static_cast<int>(scr[15]) + __builtin_popcount(scr[7]);
    if (lane < RING && lane > used_n && used_n >= 1) {
        const int last = static_cast<int>(scr[16 + (used_n < RING ? used_n : RING - 1)]);
        const int g = last + (lane - used_n);
        if (g < V) scr[16 + lane] = static_cast<uint32_t>(g);
    }
}
// ring entry of vocabulary entry g (anything for an entry the text does not use)
__device__ __forceinline__ int narrow_rank(lds_vuint* scr, int g, int blank) {
    const uint32_t word = scr[(g >> 5) & 7];
    const int r = 1 + static_cast<int>(scr[8 + ((g >> 5) & 7)]) + __builtin_popcount(word & ((1u << (g & 31)) - 1u));
    return g == blank ? 0 : r;
}
__device__ __forceinline__ int narrow_count(lds_vuint* scr) {   // labels beside the blank: narrow_max_labels(ring) fit
    return static_cast<int>(scr[15]) + __builtin_popcount(scr[7]);
}
#define CTCFA_SPIN_REPORT() do { if ((threadIdx.x & 63) == 0 && flags[20] != 0) __hip_atomic_store(fill_err, run_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)

// CTCFA_STAMP=4 (tuning builds): a TIMELINE -- every tile leaves an s_memtime stamp at the end of every 16-row group, the
// producers one per block they publish and one per wait for ring space, for the first kTraceWgs workgroups; plus the
// wave's HW_ID (which SIMD / CU it runs on).  tools/trace4.py turns them into who-waits-for-whom tables.
constexpr int kTraceWgs = 16;
constexpr int kTraceSlots = 1024;                // u64 per (workgroup, wave): [g] group end, [256 + g] at the neighbour poll, [512 + g] past it, [768 + j] block start
// (the timeline lives BEHIND the caller's char_prob buffer, 16 bytes past the word `fill_err` points at in stamp builds:
// inside the buffer the owner tiles' last-column scores would overwrite it)
#define CTCFA_TRACE_PTR (reinterpret_cast<unsigned long long*>(fill_err) + 2)
#if defined(CTCFA_STAMP) && CTCFA_STAMP == 4
#define CTCFA_TRACE(wave_slot, slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < ctcfa::kTraceWgs && (slot) < ctcfa::kTraceSlots) \
    (CTCFA_TRACE_PTR + ((int)blockIdx.x * 16 + (wave_slot)) * ctcfa::kTraceSlots)[(slot)] = __builtin_amdgcn_s_memtime(); } while (0)
#define CTCFA_TRACE_HWID(wave_slot) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < ctcfa::kTraceWgs) { unsigned hw_, xcc_; \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_)); \
    (CTCFA_TRACE_PTR + ((int)blockIdx.x * 16 + (wave_slot)) * ctcfa::kTraceSlots)[ctcfa::kTraceSlots - 1] = \
        ((unsigned long long)xcc_ << 32) | hw_; } } while (0)
#else
#define CTCFA_TRACE(wave_slot, slot) do { } while (0)
#define CTCFA_TRACE_HWID(wave_slot) do { } while (0)
#endif

__host__ __device__ constexpr int halo_lanes(int K) { return (kHaloRows + K - 1) / K; }

__device__ __forceinline__ float dpp_wave_shr1(float old_lane0, float src) {
    // lane i <- src[lane i-1]; lane 0 keeps `old_lane0` (bound_ctrl off)
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old_lane0),
                                                      __float_as_int(src), 0x138, 0xf, 0xf, false));
}

__device__ __forceinline__ float dpp_wave_shr1_zero(float src) {
    // lane i <- src[lane i-1]; lane 0 <- 0 (bound_ctrl: the DPP folds into the consuming VALU op)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x138, 0xf, 0xf, true));
}

__device__ __forceinline__ void lds_barrier() {
    // workgroup barrier that orders LDS traffic only: lowers to s_waitcnt lgkmcnt(0) +
    // s_barrier, with no vmcnt drain for global stores that nobody in this kernel reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

__device__ __forceinline__ float max3f(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);
}

// ---------------------------------------------------------------------------------------
// Fill kernel.  grid = B workgroups (one per segment), block = 64 * nwaves threads; the role
// table says which wave computes which tile, which ones are producers (stage emission rows
// global -> (e, m) pairs in the LDS ring) and which ones leave at once.
//
// Tiles do NOT hand a boundary column to each other row by row.  Column c at row t depends on
// columns c-r .. c of row t-r only, so a tile that also carries the HL*K >= 16 columns to its left
// (its halo lanes) can run 16 rows on its own: row r of a group is wrong in the first r halo
// columns and right everywhere else.  Every 16 rows (a "group") the halo lanes are refreshed from
// what the left neighbour published at the end of the same group -- ONE small LDS exchange per 16
// rows instead of a ds_read/ds_write per row, no pipeline skew between tiles (all of them work on
// the same 32-row block: the emission ring is NS = 4 slots whatever the tile count) and no
// workgroup barrier in the steady state: progress counters in LDS (done[w] = groups tile w has
// finished, staged[p] = blocks producer p has staged) are all the waves wait on --
//   tile w, kPollLead rows before the end of group g:   done[w-1] >= g + 1   (w > 0)
//   tile w, before block j:                             staged[*] >  j
//   producer, before it overwrites the slot of block j: done[*]   >= 2 (j - NS + 1)
// LDS operations of one wave execute in order, so "data, then counter" needs no wait in between.
// Roles never mix: compute waves issue only global STORES (trace words, last-column scores) and
// never wait on vmcnt; producers issue only LOADS.
// dynamic LDS = NS slots * kRows * (VP+2) * 8  +  exchange rings  +  last-column ring + counters.
// ---------------------------------------------------------------------------------------
template <int K, int VP, bool CK = false>
__global__ void __attribute__((amdgpu_waves_per_eu(((K <= 2 || (K == 3 && CK)) && VP <= 64 && CTCFA_VGPR_CAP) ? 8 : 1)))   // narrow tiles: 64 VGPRs, room for the backtrack beside it
__launch_bounds__((K >= 10) ? 320 : (K >= 8) ? 512 : 1024)
fill_kernel(const SegDesc* __restrict__ segs, const float* __restrict__ lpz,
            const int32_t* __restrict__ labels, uint32_t* __restrict__ bits,
            float* __restrict__ lastcol, int V, int blank, int cost_flags,
            const FillRoles* __restrict__ roles, const WatchDesc* __restrict__ watch,
            int32_t* __restrict__ fill_err, int run_id) {
    // cost_flags: bit 0 = preamble_transition_cost_zero (column 0 stays for free), bit 1 =
    // blank_transition_cost_zero (a column labelled blank stays for free: the blank entry's m is 0), bit 2 (32-entry
    // pitch only) = a NARROWED plan (see narrow_build): the producers stage the 32 vocabulary entries the segment's
    // text uses, the tiles address the ring by ring entry
    const bool preamble = (cost_flags & 1) != 0;
    const bool gratis = (cost_flags & 2) != 0;
    constexpr bool kRingPitch = VP == 32 || VP == 64;   // the pitches a narrowed plan runs at
    const bool narrowed = kRingPitch && CK && (cost_flags & 4) != 0;   // (narrowed plans run in checkpoint mode)
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr bool E_ALONE = VP > 80;           // the ring holds e alone (4 B an entry), the tiles work out m: see the producers
    static_assert(!(VP > 64 && CK), "checkpoint mode exists for vocabularies of at most 64 entries");
    constexpr int PITCH = E_ALONE ? VP + 4 : VP + kPitchPad;  // row pitch in entries; entry VP = start-column pseudo label
    constexpr int ROW_BYTES = PITCH * (E_ALONE ? 4 : 8);
    constexpr int SLOT_BYTES = kRows * ROW_BYTES;
    constexpr int HL = halo_lanes(K);      // halo lanes of a tile
    constexpr int XW = HL * K;             // floats per exchange row

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave_id = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveRole my = roles->wave[wave_id];
    const int my_prio = __builtin_amdgcn_readfirstlane((my.role >> 4) & 3);
    my.role = (int8_t)(my.role & 15);
    const int W = roles->nstages;         // compute tiles
    const int NS = roles->nslots;         // emission ring slots
    constexpr int XR = kExchangeRing;     // exchange ring: groups a tile can be ahead of its right neighbour (>= 2 NS; a power of two: g % XR is an AND)
    const int w = my.stage;               // this wave's tile (compute waves)

    const SegDesc sd = segs[blockIdx.x];
    if (sd.prestatus != 0 || sd.fill_skip) return;  // uniform: nothing to fill (or filled by its group's leader)
    const int T = sd.T, C = sd.C, shift = sd.shift;
    const float* __restrict__ seg_lpz = lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off;
    float* __restrict__ seg_lastcol = lastcol + sd.frm_off;
    const int rblank = narrowed ? 0 : blank;   // the blank's entry of the RING
    const int wn = (K <= kWatchMaxK) ? sd.watch_n : 0;   // watch columns of a shared fill
    const WatchDesc* __restrict__ seg_watch = watch + sd.watch_first;

    // LDS after the emission ring:
    //   xch    [W][XR][XW] floats  exchange rows: tile w's last XW columns at the end of group g in [w][g % XR]
    //   lcring [64] floats         last label column (owner tile), index t % 64
    //   flags  [32] ints           done[0..15], staged[16..17], posflag [18]
    //   sink   1 KB                target of the owner tile's lanes that publish nothing
    //   wring  [nw][64] floats     shared fills: one ring like lcring per watch column
    //   wtab   [nw] int64          ... and where its scores go in `lastcol`
    const uint32_t xch_base = static_cast<uint32_t>(NS * SLOT_BYTES);
    const uint32_t lcring_base = xch_base + static_cast<uint32_t>(W * XR * XW * 4);
    const uint32_t flag_base = lcring_base + 64 * 4;
    const uint32_t sink_base = flag_base + kFlagInts * 4;
    const uint32_t wring_base = sink_base + kSinkBytes;
    const uint32_t wtab_base = wring_base + static_cast<uint32_t>(roles->reserved[0] * 64 * 4);   // reserved[0]: watch columns of the widest group
    // (an explicit LDS pointer: a volatile access through a generic pointer compiles to flat_load +
    // s_waitcnt vmcnt(0), which would make the tiles wait for their own trace stores)
    lds_vint* flags = (lds_vint*)(smem + flag_base);
    // posflag: set by a producer as soon as any staged emission is not <= 0 -- from then on nobody
    // may assume that unreachable cells are -1e9
    lds_vint* posflag = flags + 18;
    const int nblk = (T - 1 + kRows - 1) / kRows;
    const int Cpad = roles->cpad;

    {
        float* xch = reinterpret_cast<float*>(smem + xch_base);
        for (int i = tid; i < W * XR * XW; i += blockDim.x) xch[i] = kProbMax;
        if (tid < kFlagInts) flags[tid] = ((tid == 17 && roles->nprod < 2) || tid == 19) ? kBigCount : 0;   // (19: the counter tile 0 "waits" on)
        if (tid < wn) reinterpret_cast<int64_t*>(smem + wtab_base)[tid] = seg_watch[tid].frm_off;
    }
    // a narrowed plan: what this wave needs of the segment's table, looked up before the barrier -- the table lies in the
    // emission ring's first slot, which no producer writes before it
    // producers: the vocabulary entry behind the ring entry this lane stages -- entry lane >> 1 of the 32-entry ring (fewer
    // than 32 entries: the last one again), entry `lane` of the 64-entry one
    int nar_ent = VP == 64 ? (lane < V ? lane : V - 1) : ((lane >> 1) < V ? (lane >> 1) : V - 1);
    int nar_lab[K];            // tiles: the ring entries of this lane's label columns
#pragma unroll
    for (int k = 0; k < K; ++k) nar_lab[k] = 0;
    if (kRingPitch && CK && narrowed) {
        lds_vuint* scr = (lds_vuint*)(smem + static_cast<uint32_t>(wave_id * kNarrowScratchBytes));
        int mine[K];   // (asked for before the scan of the text: one memory round trip for both)
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = my.cbase + lane * K + k - shift;
            mine[k] = (my.role != kRoleProducer && c > 0 && c < C) ? seg_lab[c] : blank;
        }
        narrow_build<VP == 64 ? 64 : 32>(scr, seg_lab, C, V, blank, lane);
        if (my.role == kRoleProducer) {
            nar_ent = static_cast<int>(scr[16 + (VP == 64 ? lane : (lane >> 1))]);
        } else {
#pragma unroll
            for (int k = 0; k < K; ++k) nar_lab[k] = narrow_rank(scr, mine[k], blank) & (VP == 64 ? 63 : 31);
        }
    }
    lds_barrier();  // the only workgroup barrier: everything after it is counter-paced
    if (my.role == kRoleIdle) return;

    if (my.role == kRoleProducer) {
        // ============================ producer wave ===========================================
        // Block jb = rows t in [32*jb + 1, 32*jb + 32] -> ring slot jb % NS.  Loads run a block
        // ahead of the LDS writes (registers), the writes up to NS - 1 blocks ahead of the tiles.
        // Little work (~5 % of a SIMD's issue slots) but everybody's critical path: without priority the
        // tiles on its SIMD starve it and the whole workgroup runs at the producer's pace.
        set_wave_prio(my_prio);
        const int part = my.stage;  // which share of the rows (nprod == 2), and which staged[] counter
        int fix_mode = 0;           // how two producers split a block: 0 every other row, 2 every other group of four rows, 3 halves
        constexpr int PASSES = kRows * VP / 64;
        constexpr int CH = PASSES < 16 ? PASSES : (VP == 32 ? 8 : 16);  // passes per chunk (VP == 32: half a block, one per producer)
        const unsigned char* lpz_bytes = reinterpret_cast<const unsigned char*>(seg_lpz);
        bool notneg = false;  // any staged emission that is not <= 0 (NaN counts)
#ifdef CTCFA_STAMP
        unsigned long long pst_space = 0, pst_t0 = __builtin_amdgcn_s_memtime(), pst_w0 = 0, pst_write = 0;
        int pst_n = 0;
#endif
        CTCFA_TRACE_HWID(14 + my.stage);
        if (CTCFA_ABL & 8) return;
        auto wait_space = [&](int jb) {   // every tile is done with the block that slot jb % NS still holds
            if (CTCFA_ABL & 2) return;
#ifdef CTCFA_STAMP
            pst_w0 = __builtin_amdgcn_s_memtime();
#endif
            CTCFA_TRACE(14 + my.stage, 120 + jb);   // (producer rows: [jb] published, [120 + jb] began to wait for space)
            if (jb < NS) return;
            const int need = kGroups * (jb - NS + 1);
            const int idx = lane < W ? lane : 0;
            int spins = 0;
            while (__builtin_amdgcn_ballot_w64(flags[idx] < need) != 0ull) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > kSpinCap) { CTCFA_SPIN_DIAG("producer", jb, need, (int)flags[idx]); break; }
            }
            asm volatile("" ::: "memory");
#ifdef CTCFA_STAMP
            pst_space += __builtin_amdgcn_s_memtime() - pst_w0;
            pst_n += spins > 0;
            pst_w0 = __builtin_amdgcn_s_memtime();
#endif
        };
        auto publish = [&](int jb) {
            // blank_transition_cost_zero: the blank entry's stay step of the rows this wave just staged becomes 0
            // (one store per block, after the rows' own stores -- LDS executes a wave's operations in order; nothing
            // in the per-row code, so nothing in the way when the flag is off)
            if (!E_ALONE && gratis) {   // (the plan refuses the flag above 64 entries)
                // the rows of the block this wave staged: all of them; every other one; every other group of four
                const int nmine = roles->nprod == 2 ? kRows / 2 : kRows;
                const int row = roles->nprod != 2 ? lane : fix_mode == 2 ? ((lane >> 2) * 2 + part) * 4 + (lane & 3)
                                : fix_mode == 3 ? part * (kRows / 2) + lane : part + 2 * lane;
                if (lane < nmine)
                    *reinterpret_cast<float*>(smem + static_cast<uint32_t>((jb % NS) * SLOT_BYTES) +
                                              static_cast<uint32_t>(row * (PITCH * 8) + rblank * 8 + 4)) = 0.0f;
            }
            if (__builtin_amdgcn_ballot_w64(notneg) != 0ull) *posflag = 1;
            asm volatile("" ::: "memory");  // data and posflag first, then the counter (LDS executes a wave's operations in order)
            if (lane == 0) flags[16 + part] = jb + 1;
            CTCFA_TRACE(14 + part, jb);
#ifdef CTCFA_STAMP
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pst_write += __builtin_amdgcn_s_memtime() - pst_w0;
            if (jb == nblk - 1 && lane == 0 && blockIdx.x < 64) {
                unsigned long long* o = reinterpret_cast<unsigned long long*>(lastcol) + (blockIdx.x * 16 + 15) * 8;
                o[0] = __builtin_amdgcn_s_memtime() - pst_t0; o[1] = pst_space; o[2] = pst_write; o[3] = pst_n; o[4] = nblk;
            }
#endif
        };
        auto load_chunk = [&](int jb, int p0, float (&e)[CH]) {
            const int t0 = jb * kRows + 1;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int idx = (p0 + q) * 64 + lane;  // element of the block
                const int sv = idx % VP;               // vocabulary entry (lane-invariant for VP <= 64)
                const int svc = sv < V ? sv : V - 1;
                int t = t0 + idx / VP;
                t = t < T ? t : T - 1;                 // rows past the end: re-read, discarded below
                const uint32_t off = static_cast<uint32_t>(t * V + svc) * 4u;
                asm volatile("global_load_dword %0, %1, %2" : "=v"(e[q]) : "v"(off), "s"(seg_lpz) : "memory");   // (inline asm: see chunk_await)
            }
        };
        // round 4: the wait for a chunk's loads, explicit (the compiler's own, in front of a block's first use, was for the
        // NEXT block's loads too): with the next chunk's CH loads in flight behind them, this chunk's are there at vmcnt(CH)
        auto chunk_await = [&](float (&e)[CH]) {
            if constexpr (CH == 8)
                asm volatile("s_waitcnt vmcnt(8)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]));
            else
#pragma unroll
                for (int q = 0; q < CH; ++q) asm volatile("s_waitcnt vmcnt(0)" : "+v"(e[q]));
        };
        auto write_chunk = [&](int jb, int p0, const float (&e)[CH]) {
            unsigned char* slot = smem + (jb % NS) * SLOT_BYTES;
            const int t0 = jb * kRows + 1;
#pragma unroll
            for (int q = 0; q < CH; ++q) {
                const int idx = (p0 + q) * 64 + lane;
                const int sv = idx % VP;
                const int r = idx / VP;
                const int t = t0 + r;
                float lb;
                if constexpr (VP == 32) {  // two rows per pass: the row's blank entry via readlane
                    const float lo = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[q]), blank));
                    const float hi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[q]), 32 + blank));
                    lb = (lane < 32) ? lo : hi;
                } else {
                    const int tc = t < T ? t : T - 1;
                    lb = *reinterpret_cast<const float*>(lpz_bytes + static_cast<uint32_t>(tc * V + blank) * 4u);
                }
                const bool valid = t < T;
                notneg |= !(e[q] <= 0.0f);
                float2* row = reinterpret_cast<float2*>(slot + r * (PITCH * 8));
                if (sv < V) row[sv] = valid ? make_float2(e[q], max3f(lb, e[q], kProbMax)) : make_float2(0.f, 0.f);
                if (sv == 0)  // start-column pseudo entry: e = -inf, m = table[t,0]'s stay step
                    row[VP] = make_float2(-__builtin_inff(),
                                          (preamble || !valid) ? 0.0f : __builtin_fmaxf(lb, kProbMax));
            }
        };
        if (VP == 32 && CTCFA_ADDTID_PRODUCER && roles->nprod == 2) {   // 32 entries, fewer (rows of V floats, gathered like a narrowed plan's), or a narrowed plan
            // ---- round 4: the 32-entry vocabulary staged ONE ROW PER STORE.  What the producer costs the tiles is the path its
            // LDS stores share with the loads of its SIMD pair (2 cycles per register dword moved: 13 per ds_write_b128, eight of
            // them per block): ds_write_addtid_b32 has no address register -- the address is M0 + offset + 4 * lane -- and moves
            // one dword per lane in 2 cycles, and a ring row IS 64 consecutive dwords: its 32 (e, m) pairs.  So lane i loads
            // entry i >> 1 of the row (one coalesced dword load, every entry twice), every lane works out m = max3(blank, e, -1e9),
            // odd lanes keep m, even lanes e, and the row leaves in one store.  More instructions per block than the 16-byte
            // path (8 per row), so two producers share a block: the upper and the lower 16 rows, two register sets each.
            fix_mode = 3;
            constexpr int NR = kRows / 2;
            const int ent = nar_ent;   // (a narrowed plan: the vocabulary entry behind ring entry lane >> 1)
            const bool odd = (lane & 1) != 0;
            uint32_t noff[8];          // a narrowed plan: byte offsets of this lane's entry in rows 0 .. 7 of a run of rows
#pragma unroll
            for (int r = 0; r < 8; ++r) noff[r] = static_cast<uint32_t>(ent + r * V) * 4u;
            // The loads are inline asm and so are the waits for them: left to the compiler, the wait in front of a block's
            // first use was vmcnt(0..3) -- for the loads of the NEXT block too, issued a moment before: a memory round trip in
            // every block of a wave the tiles wait for (trace4: the producer took 2 800 of a block's 3 400 cycles).  vmcnt
            // retires in order: with the next block's NR loads in flight behind them, this block's are there at vmcnt(NR).
            auto aload = [&](int jb, float (&e)[NR]) {
                const int t0 = jb * kRows + 1 + part * NR;
                if (V != 32) {   // (a narrowed plan, or fewer than 32 entries) rows of V entries: the row's address is wave-uniform, an SGPR pair; the lane offset one register
                    const uint32_t lane_off = static_cast<uint32_t>(ent) * 4u;
                    if (t0 + NR <= T) {   // (uniform) all 16 rows inside the segment
                        // No arithmetic between the loads (a producer gets an issue slot every ~20 cycles beside the tiles of its
                        // SIMD, vector or scalar: a 64-bit row pointer moved on by two scalar adds a row made the 16 loads take
                        // 800 cycles instead of 550, one vector add a row 1 150): eight lane offsets kept in registers, rows
                        // r and r + 8 through two row pointers a block.
                        const float* rowp0 = seg_lpz + static_cast<uint32_t>(t0 * V);
                        const float* rowp1 = rowp0 + 8 * V;
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            asm volatile("global_load_dword %0, %1, %2" : "=v"(e[r]) : "v"(noff[r]), "s"(rowp0) : "memory");
#pragma unroll
                        for (int r = 0; r < 8; ++r)
                            asm volatile("global_load_dword %0, %1, %2" : "=v"(e[8 + r]) : "v"(noff[r]), "s"(rowp1) : "memory");
                    } else {
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            int t = t0 + r;
                            t = t < T ? t : T - 1;
                            const float* rowp = seg_lpz + static_cast<uint32_t>(t * V);
                            asm volatile("global_load_dword %0, %1, %2" : "=v"(e[r]) : "v"(lane_off), "s"(rowp) : "memory");
                        }
                    }
                } else if (t0 + NR <= T) {   // (uniform) all 16 rows inside the segment: one address register, the rows as immediates (V == 32: 128 bytes a row)
                    const uint32_t off = static_cast<uint32_t>(t0 * 32 + ent) * 4u;
#pragma unroll
                    for (int r = 0; r < NR; ++r)
                        asm volatile("global_load_dword %0, %1, %2 offset:%3" : "=v"(e[r]) : "v"(off), "s"(seg_lpz), "i"(r * 128) : "memory");
                } else {
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        int t = t0 + r;
                        t = t < T ? t : T - 1;   // rows past the end repeat the last row: nothing downstream reads what the tiles make of them
                        const uint32_t off = static_cast<uint32_t>(t * 32 + ent) * 4u;
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(e[r]) : "v"(off), "s"(seg_lpz) : "memory");
                    }
                }
            };
            auto await = [&](float (&e)[NR]) {   // (the registers pass through the asm: nothing that reads them moves above it)
                static_assert(NR == 16, "the operand list below");
                asm volatile("s_waitcnt vmcnt(16)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]),
                             "+v"(e[8]), "+v"(e[9]), "+v"(e[10]), "+v"(e[11]), "+v"(e[12]), "+v"(e[13]), "+v"(e[14]), "+v"(e[15]));
            };
            auto awrite = [&](int jb, const float (&e)[NR], auto pre_tag) {
                constexpr bool PRE = decltype(pre_tag)::value;   // preamble_transition_cost_zero (the default): no per-row start-column store
                const uint32_t rowbase = static_cast<uint32_t>((jb % NS) * SLOT_BYTES + part * NR * ROW_BYTES);   // wave-uniform: M0
                // eight rows a statement: M0 is set once for them (M0 is reserved: the compiler sets it right before each use of
                // its own and keeps nothing in it -- between separate statements it may).  A producer gets an issue slot every
                // ~20 cycles beside the tiles of its SIMD: every instruction less a row is 300 of a block's 3 300 cycles.
                static_assert(NR == 16, "two halves of eight rows");
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int r = 8 * h + q;
                        const float lb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[r]), 2 * rblank));
                        notneg |= !(e[r] <= 0.0f);
                        const float m = max3f(lb, e[r], kProbMax);
                        v[q] = odd ? m : e[r];
                        if (!PRE && lane == 0)   // start-column pseudo entry of the row (under preamble_transition_cost_zero: once, below)
                            *reinterpret_cast<float2*>(smem + rowbase + static_cast<uint32_t>(r * ROW_BYTES + VP * 8)) =
                                make_float2(-__builtin_inff(), __builtin_fmaxf(lb, kProbMax));
                    }
                    asm volatile("s_mov_b32 m0, %8\n\ts_nop 0\n\t"
                                 "ds_write_addtid_b32 %0 offset:%9+%10*0\n\tds_write_addtid_b32 %1 offset:%9+%10*1\n\t"
                                 "ds_write_addtid_b32 %2 offset:%9+%10*2\n\tds_write_addtid_b32 %3 offset:%9+%10*3\n\t"
                                 "ds_write_addtid_b32 %4 offset:%9+%10*4\n\tds_write_addtid_b32 %5 offset:%9+%10*5\n\t"
                                 "ds_write_addtid_b32 %6 offset:%9+%10*6\n\tds_write_addtid_b32 %7 offset:%9+%10*7"
                                 :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]), "s"(rowbase),
                                    "i"(8 * h * ROW_BYTES), "i"(ROW_BYTES)
                                 : "memory");
                }
            };
            if (preamble && part == 0)   // start-column entry (e = -inf, stay step 0) of every row of the ring, once
                for (int idx = lane; idx < NS * kRows; idx += 64)
                    *reinterpret_cast<float2*>(smem + static_cast<uint32_t>((idx / kRows) * SLOT_BYTES + (idx % kRows) * (PITCH * 8) + VP * 8)) =
                        make_float2(-__builtin_inff(), 0.0f);
            auto arun = [&](auto pre_tag) {
                // (a next set is ALWAYS in flight behind the one waited for -- past the last block its loads are issued once more,
                // 16 x 256 bytes a segment -- so that one wait, vmcnt(16), serves every block)
                float ea[NR], eb[NR];
                aload(0, ea);
                for (int jb = 0; jb < nblk; jb += 2) {
                    aload(jb + 1 < nblk ? jb + 1 : nblk - 1, eb);
                    wait_space(jb);
                    await(ea);
                    awrite(jb, ea, pre_tag);
                    publish(jb);
                    if (jb + 1 >= nblk) break;
                    aload(jb + 2 < nblk ? jb + 2 : nblk - 1, ea);
                    wait_space(jb + 1);
                    await(eb);
                    awrite(jb + 1, eb, pre_tag);
                    publish(jb + 1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the set nobody uses: its registers must not be handed on with a load pending)
            };
            if (preamble) arun(std::true_type{});
            else arun(std::false_type{});
        } else if ((VP == 32 || VP == 64) && V == VP) {  // (constant-folded away for the other pitches)
            // ---- vectorised staging (the common case V == 32): a lane loads 4 consecutive
            // entries of a row with one dwordx4, LPR lanes share a row, RPP rows per pass.
            constexpr int LPR = VP / 4;
            constexpr int RPP = 64 / LPR;
            constexpr int NPB = (VP <= 64) ? kRows / RPP : 1;  // 4 (VP=32) or 8 (VP=64) passes per block
            // 64 entries: two producers, each takes every other pass (four rows) of a block -- four passes and
            // two register sets per wave, like the single producer of the 32-entry case (one producer with
            // one set had a block's HBM latency in front of every block: 207 us)
            constexpr int PSTEP = (VP == 64 || (VP == 32 && CTCFA_TWO_PROD32)) ? 2 : 1;   // (two producers for the 32-entry case too: measured in round 3, 143.8 against 143.1 us)
            constexpr int NP = NPB / PSTEP;
            const int lr = lane / LPR;
            const int lv = (lane % LPR) * 4;
            const int blank_lane = lr * LPR + blank / 4;  // lane of my row that holds the blank entry
            const int blank_comp = blank & 3;
            auto vload = [&](int jb, float4 (&e)[NP]) {
                const int t0 = jb * kRows + 1 + lr;
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int p = pp * PSTEP + (PSTEP == 2 ? part : 0);
                    int t = t0 + p * RPP;
                    t = t < T ? t : T - 1;
                    e[pp] = *reinterpret_cast<const float4*>(lpz_bytes + static_cast<uint32_t>(t * V + lv) * 4u);
                }
            };
            auto vwrite = [&](int jb, const float4 (&e)[NP]) {
                const uint32_t slot = static_cast<uint32_t>((jb % NS) * SLOT_BYTES);
                const uint32_t ent = slot + static_cast<uint32_t>((lr * PITCH + lv) * 8);
                // start-column entry: written by the lane holding entries 0..3, others hit a sink
                const uint32_t spc = (lv == 0) ? slot + static_cast<uint32_t>((lr * PITCH + VP) * 8)
                                               : sink_base + static_cast<uint32_t>(lane * 8);
                const int t0 = jb * kRows + 1 + lr;
                const bool paced = jb >= 2 * NS;   // (the first rings' worth goes out as fast as it can: the tiles are waiting for it)
#pragma unroll
                for (int pp = 0; pp < NP; ++pp) {
                    const int p = pp * PSTEP + (PSTEP == 2 ? part : 0);
                    const float4 v = e[pp];
                    notneg |= !(max3f(v.x, v.y, __builtin_fmaxf(v.z, v.w)) <= 0.0f) | (v.x != v.x) | (v.y != v.y) | (v.z != v.z) | (v.w != v.w);
                    const float mine = blank_comp == 0 ? v.x : (blank_comp == 1 ? v.y : (blank_comp == 2 ? v.z : v.w));
                    const float lb = __shfl(mine, blank_lane);
                    const bool valid = (t0 + p * RPP) < T;
                    float4 lo = make_float4(v.x, max3f(lb, v.x, kProbMax), v.y, max3f(lb, v.y, kProbMax));
                    float4 hi = make_float4(v.z, max3f(lb, v.z, kProbMax), v.w, max3f(lb, v.w, kProbMax));
                    if (!valid) { lo = make_float4(0.f, 0.f, 0.f, 0.f); hi = lo; }
                    const float2 sp = make_float2(-__builtin_inff(),
                                                  (preamble || !valid) ? 0.0f : __builtin_fmaxf(lb, kProbMax));
                    unsigned char* q = smem + ent + p * (RPP * PITCH * 8);
                    // Round 4 (the ablations of profiles/r04_fill_ablations.txt): these few stores are what the producer costs the
                    // tiles -- 17 us of the 140 -- not its loads or its vector work.  A store moves its address and data registers to
                    // the LDS over a path the wave shares with every other wave of its SIMD pair (2 cycles a dword: 13 per
                    // ds_write_b128), and a block's twelve in a row hold that path longer than the tiles' two rows of prefetched
                    // gathers last: every tile of the pair stood still once per block.  In the steady state the producer is a ring
                    // ahead and has a block's time for its stores: it sleeps between them (s_sleep 4 = 256 clocks), so that each is
                    // a hiccup the tiles' prefetch absorbs; and the start-column entry, a constant under
                    // preamble_transition_cost_zero, is written once for the whole ring instead of once per row.
                    if (CTCFA_ABL & 32) { lo = make_float4(v.x, v.x, v.y, v.y); hi = make_float4(v.z, v.z, v.w, v.w); }   // (tuning: no vector work on the values)
                    if (CTCFA_ABL & 16) { asm volatile("" :: "v"(lo.x), "v"(lo.y), "v"(lo.z), "v"(lo.w), "v"(hi.x), "v"(hi.y), "v"(hi.z), "v"(hi.w), "v"(sp.y)); continue; }   // (tuning: no LDS stores)
                    if (CTCFA_ABL & 64) {   // (tuning: the same bytes as four 8-byte stores instead of two 16-byte ones)
                        *reinterpret_cast<float2*>(q) = make_float2(lo.x, lo.y);
                        *reinterpret_cast<float2*>(q + 8) = make_float2(lo.z, lo.w);
                        *reinterpret_cast<float2*>(q + 16) = make_float2(hi.x, hi.y);
                        *reinterpret_cast<float2*>(q + 24) = make_float2(hi.z, hi.w);
                    } else {
                    *reinterpret_cast<float4*>(q) = lo;
#if CTCFA_PROD_PACE > 0
                    if (paced) __builtin_amdgcn_s_sleep(CTCFA_PROD_PACE);
#endif
                    *reinterpret_cast<float4*>(q + 16) = hi;
#if CTCFA_PROD_PACE > 0
                    if (paced) __builtin_amdgcn_s_sleep(CTCFA_PROD_PACE);
#endif
                    }
                    // sink offsets stay inside the 1 KB sink: p * RPP * PITCH * 8 would not
                    if (!preamble && !(CTCFA_ABL & 128))   // (under preamble_transition_cost_zero: written once, below)
                    *reinterpret_cast<float2*>(smem + spc + ((lv == 0) ? p * (RPP * PITCH * 8) : 0)) = sp;
                }
            };
            if constexpr (PSTEP == 2) fix_mode = 2;
            if (preamble && part == 0)   // start-column entry (e = -inf, stay step 0) of every row of the ring, once (before this wave's first publish: LDS order)
                for (int idx = lane; idx < NS * kRows; idx += 64)
                    *reinterpret_cast<float2*>(smem + static_cast<uint32_t>((idx / kRows) * SLOT_BYTES + (idx % kRows) * (PITCH * 8) + VP * 8)) =
                        make_float2(-__builtin_inff(), 0.0f);
            {   // two register sets: loads run a block ahead of the LDS writes
                // (a third set -- loads two blocks ahead -- measured no faster, and does not fit the
                // 64-register budget of the narrow tiles)
                float4 ea[NP], eb[NP];
                vload(0, ea);
                for (int jb = 0; jb < nblk; jb += 2) {
                    if (jb + 1 < nblk) vload(jb + 1, eb);
                    wait_space(jb);
                    vwrite(jb, ea);
                    publish(jb);
                    if (jb + 1 >= nblk) break;
                    if (jb + 2 < nblk) vload(jb + 2, ea);
                    wait_space(jb + 1);
                    vwrite(jb + 1, eb);
                    publish(jb + 1);
                }
            }
        } else if constexpr (VP > 32 && VP <= 64) {
            // ---- one row per pass (character vocabularies between 33 and 64 entries that are not
            // a compiled pitch themselves, e.g. the reference's 38-token model): lane = vocabulary
            // entry, the row's address is wave-uniform (scalar arithmetic), the blank entry comes
            // from a v_readlane, LDS offsets are immediates.  Lane VP writes the start-column pseudo
            // entry, the remaining lanes a sink.  ~6 instructions per row.
            constexpr int kPseudoLane = VP < 64 ? VP : 63;  // V < VP here, so this lane carries no entry
            // lanes past the vocabulary re-read its last entry; a narrowed plan's 64-entry ring: the vocabulary entry behind ring entry `lane`
            const int svl = VP == 64 ? nar_ent : (lane < V ? lane : V - 1);
            // LDS entry a lane writes: its own, the pseudo entry VP, or (lanes past VP) the pad entry VP+1
            const uint32_t ent = static_cast<uint32_t>((lane == kPseudoLane ? VP : (lane <= VP ? lane : VP + 1)) * 8);
            // With two producers (roles->nprod == 2) each stages every other row of a block and
            // counts its blocks in its own staged[] entry.
            auto run = [&](auto parts_tag) {
                constexpr int PARTS = decltype(parts_tag)::value;
                constexpr int NR = kRows / PARTS;  // rows of a block this wave stages: part, part + PARTS, ..
                // (round 4: loads and the wait for them in inline asm, as in the 32-entry producer above -- the compiler's own wait
                // in front of a block's first use was for the NEXT block's loads too; the row address is wave-uniform: an SGPR pair)
                const uint32_t lane_off = static_cast<uint32_t>(svl) * 4u;
                auto rload = [&](int jb, float (&e)[NR]) {
                    const int t0 = jb * kRows + 1 + part;
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        int t = t0 + i * PARTS;
                        t = t < T ? t : T - 1;
                        const float* rowp = seg_lpz + static_cast<uint32_t>(t * V);
                        asm volatile("global_load_dword %0, %1, %2" : "=v"(e[i]) : "v"(lane_off), "s"(rowp) : "memory");
                    }
                };
                auto rawait = [&](float (&e)[NR]) {   // this wave's set of NR loads is there once only the NR issued after it are in flight
                    if constexpr (NR == 16)
                        asm volatile("s_waitcnt vmcnt(16)" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]), "+v"(e[4]), "+v"(e[5]), "+v"(e[6]), "+v"(e[7]),
                                     "+v"(e[8]), "+v"(e[9]), "+v"(e[10]), "+v"(e[11]), "+v"(e[12]), "+v"(e[13]), "+v"(e[14]), "+v"(e[15]));
                    else
#pragma unroll
                        for (int i = 0; i < NR; ++i) asm volatile("s_waitcnt vmcnt(0)" : "+v"(e[i]));
                };
                auto rwrite = [&](int jb, const float (&e)[NR]) {
                    unsigned char* dst = smem + static_cast<uint32_t>((jb % NS) * SLOT_BYTES) + ent +
                                         static_cast<uint32_t>(part * (PITCH * 8));
#pragma unroll
                    for (int i = 0; i < NR; ++i) {
                        // (rows past the end of the segment repeat its last row -- the loads are clamped: nothing
                        // downstream reads what the tiles make of them, so they are not zeroed here)
                        const float lb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e[i]), rblank));
                        notneg |= !(e[i] <= 0.0f);
                        float2 v = make_float2(e[i], max3f(lb, e[i], kProbMax));
                        if (lane == kPseudoLane) v = make_float2(-__builtin_inff(), preamble ? 0.0f : __builtin_fmaxf(lb, kProbMax));
                        *reinterpret_cast<float2*>(dst + i * (PARTS * PITCH * 8)) = v;
                    }
                };
                if constexpr (PARTS == 2) {
                    // two register sets of 16 rows: the loads of a block are issued a whole block before
                    // they are written (HBM latency never sits inside the staging of a block)
                    // (a next set is always in flight behind the one waited for: past the last block its loads are issued again)
                    float ea[NR], eb[NR];
                    rload(0, ea);
                    for (int jb = 0; jb < nblk; jb += 2) {
                        rload(jb + 1 < nblk ? jb + 1 : nblk - 1, eb);
                        wait_space(jb);
                        rawait(ea);
                        rwrite(jb, ea);
                        publish(jb);
                        if (jb + 1 >= nblk) break;
                        rload(jb + 2 < nblk ? jb + 2 : nblk - 1, ea);
                        wait_space(jb + 1);
                        rawait(eb);
                        rwrite(jb + 1, eb);
                        publish(jb + 1);
                    }
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the set nobody uses)
                } else {
                    // One register set: the loads of block jb+1 are issued right after block jb has been written.
                    float ea[NR];
                    rload(0, ea);
                    for (int jb = 0; jb < nblk; ++jb) {
                        wait_space(jb);
                        rawait(ea);
                        rwrite(jb, ea);
                        publish(jb);
                        if (jb + 1 < nblk) rload(jb + 1, ea);
                    }
                }
            };
            // (Tried in round 2 and dropped: dwordx2 loads with two or three rows per pass for even vocabularies --
            // fewer passes, but lane-varying row addresses and store targets; V = 38: 183 us against 166.)
            run(std::integral_constant<int, 2>{});   // (the plan always gives this path two producers)
        } else if constexpr (VP == 32) {
            // V < 32 (e.g. the 29..31 entries of English character models): a lane loads one entry, two rows per
            // pass; two producers, each stages one half (16 rows, 8 passes) of every block with two register sets
            // -- the loads of a block are issued a whole block before they are written.  (One producer with one
            // set of 16 had a block's HBM latency in front of every block: 212 us at V = 20; two sets of 16 in
            // one wave do not fit the 64 registers the narrow tiles are held to.)
            fix_mode = 3;
            const int p0 = part * CH;
            float ea[CH], eb[CH];
            load_chunk(0, p0, ea);
            for (int jb = 0; jb < nblk; jb += 2) {
                load_chunk(jb + 1 < nblk ? jb + 1 : nblk - 1, p0, eb);   // (past the last block: its loads once more, so that one wait serves every block)
                wait_space(jb);
                chunk_await(ea);
                write_chunk(jb, p0, ea);
                publish(jb);
                if (jb + 1 >= nblk) break;
                load_chunk(jb + 2 < nblk ? jb + 2 : nblk - 1, p0, ea);
                wait_space(jb + 1);
                chunk_await(eb);
                write_chunk(jb + 1, p0, eb);
                publish(jb + 1);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the set nobody uses)
        } else if constexpr (VP > 64 && !E_ALONE) {
            // ---- character vocabularies between 65 and 80 entries (e.g. 76 for French), (e, m) pairs as below 64: a row per
            // pass, lane i holds entries i and 64 + i; lane 0 also writes the start-column pseudo entry.
            // Two producers, each stages every other row.  Under preamble_transition_cost_zero the start-column entry
            // is written once for the whole ring; rows past the end of the segment repeat its last row (nothing
            // reads what follows).
            const int sv0 = lane < V ? lane : V - 1;
            const int sv1 = lane + 64 < V ? lane + 64 : V - 1;
            auto runw = [&](auto parts_tag) {
                constexpr int PARTS = decltype(parts_tag)::value;
                constexpr int NR = kRows / PARTS;
                auto wload = [&](int jb, float (&e0)[NR], float (&e1)[NR]) {
                    const int t0 = jb * kRows + 1 + part;
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        int t = t0 + r * PARTS;
                        t = t < T ? t : T - 1;
                        const unsigned char* rowp = lpz_bytes + static_cast<uint32_t>(t * V) * 4u;
                        e0[r] = *reinterpret_cast<const float*>(rowp + static_cast<uint32_t>(sv0) * 4u);
                        e1[r] = *reinterpret_cast<const float*>(rowp + static_cast<uint32_t>(sv1) * 4u);
                    }
                };
                auto wwrite = [&](int jb, const float (&e0)[NR], const float (&e1)[NR]) {
                    unsigned char* slot = smem + static_cast<uint32_t>((jb % NS) * SLOT_BYTES) + static_cast<uint32_t>(part * (PITCH * 8));
                    // lanes past the vocabulary park their stores on the pad entry VP + 1
                    unsigned char* d0 = slot + static_cast<uint32_t>((lane < V ? lane : VP + 1) * 8);
                    unsigned char* d1 = slot + static_cast<uint32_t>((lane + 64 < V ? lane + 64 : VP + 1) * 8);
                    unsigned char* dp = slot + static_cast<uint32_t>((lane == 0 ? VP : VP + 1) * 8);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const float lb = (blank < 64)
                                             ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e0[r]), blank & 63))
                                             : __int_as_float(__builtin_amdgcn_readlane(__float_as_int(e1[r]), blank & 63));
                        notneg |= !(e0[r] <= 0.0f) | !(e1[r] <= 0.0f);
                        *reinterpret_cast<float2*>(d0 + r * (PARTS * PITCH * 8)) = make_float2(e0[r], max3f(lb, e0[r], kProbMax));
                        *reinterpret_cast<float2*>(d1 + r * (PARTS * PITCH * 8)) = make_float2(e1[r], max3f(lb, e1[r], kProbMax));
                        if (!preamble)   // (uniform)
                            *reinterpret_cast<float2*>(dp + r * (PARTS * PITCH * 8)) = make_float2(-__builtin_inff(), __builtin_fmaxf(lb, kProbMax));
                    }
                };
                // One register set per producer (a second one -- 64 more registers -- spills under the 128 a
                // 16-wave workgroup may use): the loads of block jb+1 are issued right after block jb has been written.
                float ea[NR], eb[NR];
                wload(0, ea, eb);
                for (int jb = 0; jb < nblk; ++jb) {
                    wait_space(jb);
                    wwrite(jb, ea, eb);
                    publish(jb);
                    if (jb + 1 < nblk) wload(jb + 1, ea, eb);
                }
            };
            if (preamble && part == 0)   // start-column entry of every row of the ring, once
                for (int idx = lane; idx < NS * kRows; idx += 64)
                    *reinterpret_cast<float2*>(smem + static_cast<uint32_t>((idx / kRows) * SLOT_BYTES + (idx % kRows) * (PITCH * 8) + VP * 8)) =
                        make_float2(-__builtin_inff(), 0.0f);
            runw(std::integral_constant<int, 2>{});   // (the plan always gives these vocabularies two producers)
        } else if constexpr (E_ALONE) {
            // ---- vocabularies of 81 .. 256 entries (character sets like French's 76, small sub-word models): the ring
            // holds the emissions alone, one float per entry (e only: a row of (e, m) pairs is 2 x 8 B x VP -- random
            // labels then meet in the same LDS banks three or four deep, and one workgroup takes most of a CU's LDS);
            // the tiles work m = max(blank, e, -1e9) out themselves from the row's blank posterior, which lane 0 leaves
            // in entries VP + 2 (for label columns) and VP + 3 (for the start column and the padding left of it: 0 under
            // preamble_transition_cost_zero).  A row per pass, lane i holds entries i, 64 + i, ...; two producers, each
            // stages every other row.  The start-column pseudo entry (e = -inf) is written once for the whole ring; rows
            // past the end of the segment repeat its last row (nothing reads what follows).
            constexpr int NE = (VP + 63) / 64;
            int sv[NE];
#pragma unroll
            for (int q = 0; q < NE; ++q) sv[q] = lane + 64 * q < V ? lane + 64 * q : V - 1;
            auto runw = [&](auto parts_tag) {
                constexpr int PARTS = decltype(parts_tag)::value;
                constexpr int NR = kRows / PARTS;             // rows of a block this wave stages: part, part + PARTS, ...
                // Two register sets: the loads of a chunk are issued a whole chunk before they are written (with one set
                // every block had an HBM round trip in front of it: 3.3 us a block, 310 us for config 3's shape whatever
                // the vocabulary).  A chunk is this wave's 16 rows of a block, or 8 of them where a lane holds three or
                // four entries (two sets of 16 would be 96 / 128 registers).
                constexpr int CR = NE <= 2 ? NR : NR / 2;     // rows per chunk
                constexpr int CPB = NR / CR;                  // chunks per block
                auto cload = [&](int ch, float (&e)[NE][CR]) {
                    const int t0 = (ch / CPB) * kRows + 1 + part + (ch % CPB) * (CR * PARTS);
#pragma unroll
                    for (int r = 0; r < CR; ++r) {
                        int t = t0 + r * PARTS;
                        t = t < T ? t : T - 1;
                        const unsigned char* rowp = lpz_bytes + static_cast<uint32_t>(t * V) * 4u;
#pragma unroll
                        for (int q = 0; q < NE; ++q) e[q][r] = *reinterpret_cast<const float*>(rowp + static_cast<uint32_t>(sv[q]) * 4u);
                    }
                };
                auto cwrite = [&](int ch, const float (&e)[NE][CR]) {
                    unsigned char* slot = smem + static_cast<uint32_t>(((ch / CPB) % NS) * SLOT_BYTES) +
                                          static_cast<uint32_t>((part + (ch % CPB) * (CR * PARTS)) * ROW_BYTES);
                    // lanes past the vocabulary park their stores on the pad entry VP + 1
                    unsigned char* d[NE];
#pragma unroll
                    for (int q = 0; q < NE; ++q) d[q] = slot + static_cast<uint32_t>((lane + 64 * q < V ? lane + 64 * q : VP + 1) * 4);
                    unsigned char* dx = slot + static_cast<uint32_t>((VP + 2) * 4);
#pragma unroll
                    for (int r = 0; r < CR; ++r) {
                        float src = e[0][r];   // the register that holds the blank entry (wave-uniform choice)
#pragma unroll
                        for (int q = 1; q < NE; ++q) src = (blank >> 6) == q ? e[q][r] : src;
                        const float lb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(src), blank & 63));
#pragma unroll
                        for (int q = 0; q < NE; ++q) {
                            notneg |= !(e[q][r] <= 0.0f);
                            *reinterpret_cast<float*>(d[q] + r * (PARTS * ROW_BYTES)) = e[q][r];
                        }
                        if (lane == 0) *reinterpret_cast<float2*>(dx + r * (PARTS * ROW_BYTES)) = make_float2(lb, preamble ? 0.0f : lb);
                    }
                };
                const int nch = nblk * CPB;
                float ea[NE][CR], eb[NE][CR];
                cload(0, ea);
                for (int ch = 0; ch < nch; ch += 2) {
                    if (ch + 1 < nch) cload(ch + 1, eb);
                    if (ch % CPB == 0) wait_space(ch / CPB);
                    cwrite(ch, ea);
                    if (ch % CPB == CPB - 1) publish(ch / CPB);
                    if (ch + 1 >= nch) break;
                    if (ch + 2 < nch) cload(ch + 2, ea);
                    if ((ch + 1) % CPB == 0) wait_space((ch + 1) / CPB);
                    cwrite(ch + 1, eb);
                    if ((ch + 1) % CPB == CPB - 1) publish((ch + 1) / CPB);
                }
            };
            if (part == 0)   // start-column entry of every row of the ring, once (before the first publish of this wave: LDS order)
                for (int idx = lane; idx < NS * kRows; idx += 64)
                    *reinterpret_cast<float*>(smem + static_cast<uint32_t>((idx / kRows) * SLOT_BYTES + (idx % kRows) * ROW_BYTES + VP * 4)) =
                        -__builtin_inff();
            runw(std::integral_constant<int, 2>{});   // (the plan always gives these vocabularies two producers)
        }
        CTCFA_SPIN_REPORT();
        return;
    }

    // ================================ compute tiles ============================================
    const int cbase = my.cbase;
    float prev[K];
    float hx[K];        // halo values to put in at the start of the next group
    uint32_t dec[K];
    uint32_t gaddr[K];  // LDS byte address of this column's (e, m) pair (E_ALONE: of its e) in row 0 of the current slot
    bool startlike[K];  // E_ALONE: the start column or padding left of it (its m comes from entry VP + 3 of the row)
    uint32_t xaddr = static_cast<uint32_t>((VP + 2) * 4);   // E_ALONE: the row's (blank, start-column stay step) pair, same bookkeeping as gaddr
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int pc = cbase + lane * K + k;
        const int c = pc - shift;
        int lab;
        if (c <= 0) lab = VP;                  // start column / left padding
        else if (c < C) lab = narrowed ? nar_lab[k] : seg_lab[c];   // (a narrowed plan: the label's entry of the ring)
        else lab = rblank;                     // right padding: any valid entry
        gaddr[k] = static_cast<uint32_t>(lab) * (E_ALONE ? 4u : 8u);
        startlike[k] = c <= 0;
        prev[k] = (c <= 0) ? 0.0f : kProbMax;  // table[0,0] = 0, table[0,c>0] = -1e9
        hx[k] = prev[k];
        dec[k] = 0u;
    }
    const bool is_halo = (w > 0) && (lane < HL);   // tile 0's first lanes are left padding: they reproduce table[t,0] themselves
    const int wstar = sd.owner_stage;  // tile that owns the last label column (ragged batches: <= W-1)
    const int lstar = sd.owner_lane;
    bool ring_has_row0 = false;   // owner tiles: entry 0 of the current ring half holds the row the block before ended in
    // Shared fill: the watch columns this tile owns (halo lanes hold copies, not columns) are slots
    // [ws_lo, ws_hi) of the group's list; the lane that holds slot `myslot` has it at k == ksel.
    int ws_lo = 0, ws_hi = 0, myslot = -1, ksel = K - 1;
    int next_watch = 0x7fffffff;   // smallest watch column at or right of this tile's first own column
    if constexpr (K <= kWatchMaxK) {
        for (int s = 0; s < wn; ++s) {
            const int rel = seg_watch[s].pcol - cbase;
            if (rel < HL * K) continue;
            if (next_watch == 0x7fffffff) next_watch = seg_watch[s].pcol;
            if (rel >= 64 * K) break;
            if (ws_hi == 0) ws_lo = s;
            ws_hi = s + 1;
            if (lane == rel / K) {
                myslot = s;
                ksel = rel % K;
            }
        }
    }
    const bool watch_tile = ws_hi > ws_lo;
    // exchange rows: what I read (left neighbour's ring; lanes past the halo re-read its last entry) and
    // what I publish (my last HL lanes)
    const uint32_t xin_addr = xch_base + static_cast<uint32_t>(((w > 0 ? w - 1 : 0) * XR * XW + (lane < HL ? lane : HL - 1) * K) * 4);
    const uint32_t xout_addr = xch_base + static_cast<uint32_t>((w * XR * XW + (lane >= 64 - HL ? lane - (64 - HL) : 0) * K) * 4);
    const bool publishes = lane >= 64 - HL;
    // K <= 2: at a group's end EVERY lane stores (no branch, no exec-mask juggling): the lanes that publish nothing write into
    // the sink (their own 4 K bytes of it per exchange slot: lane * 4 K + 8 slots * 4 XW <= 1 KB), and every lane but the last
    // stores its copy of the counter there too (the last 256 bytes of the sink)
    constexpr bool kLean = (VP == 32 || (CTCFA_LEAN_ALL_PITCHES && VP <= 64)) && CTCFA_LEAN_HANDOVER;   // (the other pitches: rounds 2-3's hand-over, below)
    constexpr bool kAllLanesPublish = K == 1 || (K == 2 && !CTCFA_MASKED_PUBLISH);
    const uint32_t xw_addr = (!kAllLanesPublish || publishes) ? xout_addr : sink_base + static_cast<uint32_t>(lane * K * 4);
    const uint32_t cnt_out_addr = (lane == 63) ? flag_base + static_cast<uint32_t>(w * 4) : sink_base + 1024u + static_cast<uint32_t>(lane * 4);

    // Tiles of one SIMD compete for issue slots by priority, then age: the later-dispatched waves
    // would always lose.  A tile is only ever waited for by its right neighbour, so the left ones go first.
    // Shipped: the later half of the tiles first (measured on config 3, round 2: earlier tiles first 160 us, later
    // half first 148, producer above the tiles 165-170, "a tile that had to wait steps back" 155).  Round 3: tiles
    // 3 / 2, producers 1 -- one above the striders of the previous batch's backtrack (priority 0), which otherwise
    // share the producers' level: 0.1559 -> 0.1542 ms per pipelined step, the fill alone unchanged.
    set_wave_prio(my_prio);   // (the role table: ctcfa.hip tile_roles -- later half CTCFA_TILE_PRIO_BASE + 1, earlier half CTCFA_TILE_PRIO_BASE)

    // Dead zone: column c cannot reach the end cell's column C-1 from rows t > T-C+c, so the
    // backtrack never visits those cells and they feed only other dead cells (in this tile or, through
    // the exchange rows, in the halo of the next one).  A tile stops after the last block in which
    // its right-most column is still alive (exact for any input).
    int jlast = nblk - 1;
    {
        const int cmax = (cbase + 64 * K - 1) - shift;     // right-most column of this tile
        // Shared fill: a column serves every member whose last column lies at or right of it; the
        // shortest of them keeps it alive longest (and a watch column is needed in every row).
        const int cend = (wn > 0 && next_watch != 0x7fffffff) ? next_watch - shift : C - 1;
        if (cmax < cend) {
            const int tdead = T - (cend + 1) + cmax;       // last row where cmax is alive
            jlast = tdead >= 1 ? (tdead - 1) / kRows : -1;
            if (jlast > nblk - 1) jlast = nblk - 1;
        }
    }
    if (w > wstar) jlast = -1;  // ragged batch: this tile holds right padding only
    // Unreachable zone: cells with c > t hold exactly -1e9 as long as every emission so far is
    // <= 0 (log-probabilities always are; the producers' posflag says when they are not).  A tile
    // whose left-most column (halo included) is cmin idles through the blocks that end before row
    // cmin and then starts from the state it would have computed: -1e9 in every column.  Its exchange
    // rows still hold the -1e9 they were initialised with, so it only counts the groups.
    int jfirst = 0;
    {
        const int cmin = cbase - shift;                    // left-most column of this tile
        if (cmin >= 1) jfirst = (cmin - 1) / kRows;
        if (jfirst > jlast + 1) jfirst = jlast + 1;
    }
#ifdef CTCFA_NO_DEADZONE
    jlast = nblk - 1;
    jfirst = 0;
#endif

    auto staged_now = [&]() -> int {
        const int a = flags[16], b = flags[17];
        return __builtin_amdgcn_readfirstlane(a < b ? a : b);
    };
    int staged_seen = 0;
    int peek_sa = 0, peek_sb = 0;   // counter values on their way from LDS
#ifdef CTCFA_STAMP   // tuning builds: where a tile's cycles go (tools/stamps2.py reads them from the lastcol workspace)
    unsigned long long st_nbr = 0, st_staged = 0, st_t0 = __builtin_amdgcn_s_memtime();
    int st_nbr_n = 0, st_staged_n = 0;
    unsigned long long st_first[2] = {0, 0};
#define CTCFA_STAMP_BEGIN() const unsigned long long st_a = __builtin_amdgcn_s_memtime()
#define CTCFA_STAMP_END(acc, cnt) do { acc += __builtin_amdgcn_s_memtime() - st_a; ++cnt; } while (0)
#else
#define CTCFA_STAMP_BEGIN() do { } while (0)
#define CTCFA_STAMP_END(acc, cnt) do { } while (0)
#endif
    CTCFA_TRACE_HWID(w);
    int cur_slot = 0;  // slot whose offset is folded into gaddr[]
    int jslot = 0;     // j % NS of the block loop below

    // scores of the watch columns of this tile, rows 32j .. 32j+31 (from the rings, or -1e9 for a
    // block the tile skipped), to where the members' backtracks look for them
    auto watch_out = [&](int j, bool from_ring) {
        const int t = j * kRows + (lane & 31);
        for (int s = ws_lo + (lane >> 5); s < ws_hi; s += 2) {
            const int64_t off = *reinterpret_cast<const int64_t*>(smem + wtab_base + s * 8);
            const float v = from_ring ? *reinterpret_cast<const float*>(smem + wring_base + (s * 64 + (j & 1) * kRows + (lane & 31)) * 4)
                                      : kProbMax;
            if (t >= 1 && t < T) lastcol[off + t] = v;
        }
    };
    // One BODY of the tile loop = NB consecutive 32-row blocks (NB = 2 where the ring has an even number of slots and the
    // body starts in an even one: the two slots are contiguous in LDS, every row of the 64 is `VGPR + immediate`).  Round 4:
    // what lies between two bodies -- loop latch, staged-counter check, slot arithmetic, dispatch on the tile's kind, the LDS
    // prefetch queue running dry and filling up again -- cost a tile 600..800 cycles per 32 rows (tools/trace4.py: a fifth of
    // a block's time), and the group hand-over 25 instructions; now one boundary per 64 rows and 10 instructions per group:
    //   * 4 rows before a group's end: the neighbour's counter, THEN its exchange row of this group (LDS executes a wave's
    //     operations in order; the writer stores data, then counter) -- both reads issued back to back, nothing looked at yet;
    //   * at the group's end: my own columns and my counter go out -- every lane stores (K <= 2: the lanes that publish
    //     nothing write into the sink: no branch, no exec-mask juggling); then the counter read above is compared with the
    //     group number: if the neighbour had not finished the group when the counter was read (it normally is 4+ rows
    //     ahead), the tile spins on the counter and reads the row again;
    //   * tile 0 reads a counter that is always "done" (flags[19]) and its own ring: no `w > 0` branches in the row loop.
    // OWNER 1: this tile holds the last label column and also publishes its scores; OWNER 2 (shared fill): it holds watch
    // columns, each in some lane at some k.
    lds_vint* cnt_in = flags + (w > 0 ? w - 1 : 19);
    // The owner of the last label column is the tile everybody ends up waiting for (through the ring: trace4), and what it does
    // more than the others sat on its critical path: the row's score went to the LDS ring right behind the v_max3 that makes
    // it (the wave stood still until the result was there, then for the store's register transfer: +18 cycles a row), and at
    // every block's end the 32 scores were read back and stored to HBM with a full LDS round trip in between.  Round 4: the
    // score of row i is stored while row i + 1 is computed (`own_pend`), and block j's scores leave for HBM in the middle of
    // block j + 1 (`own_copy`), from the ring half nobody writes to then.
    float own_pend = 0.0f;     // OWNER 1: last-column score of the row before, not yet in the ring
    bool own_have = false;     // ... there is one
    bool own_copy = false;     // OWNER 1: the block before this one is complete in the ring and not yet in `lastcol`
    auto block_impl = [&](int j, auto owner_tag, auto start_tag, auto nb_tag) {
        constexpr int OWNER = decltype(owner_tag)::value;
        constexpr bool START = decltype(start_tag)::value;   // E_ALONE: some column of this tile is the start column or padding left of it
        constexpr int NB = decltype(nb_tag)::value;
        constexpr int ROWS = NB * kRows;
        CTCFA_TRACE(w, 768 + j);
        const int slot = jslot;   // j % NS, kept by the block loop (NS is 3 or 4: a division otherwise)
        const uint32_t delta = static_cast<uint32_t>((slot - cur_slot) * SLOT_BYTES);
        cur_slot = slot;
#pragma unroll
        for (int k = 0; k < K; ++k) gaddr[k] += delta;
        if constexpr (E_ALONE) xaddr += delta;
        // exchange slots of this body's groups: (2j + q) % XR, q = 0 .. 2 NB - 1 (no wrap inside a body: XR is a multiple of 2 NB
        // and a two-block body starts at an even block)
        const int g0 = j * kGroups;
        const uint32_t xoff = static_cast<uint32_t>((g0 & (XR - 1)) * XW * 4);
        const uint32_t xin_cur = xin_addr + xoff, xw_cur = xw_addr + xoff;
        // Owner tiles publish the score of their watched column after every row: ring entry q of the half (j & 1)
        // holds table row 32 j + q, so row i goes to entry i + 1 and the block's last row to entry 0 of the OTHER
        // half.  One ds_write_b32 per row straight from the register (every lane stores; the lanes that watch
        // nothing into the sink) -- no vector instruction, where collecting four rows for a wider store cost a v_mov
        // per row on the tile everybody else ends up waiting for.
        uint32_t out_addr = sink_base + static_cast<uint32_t>(lane * 4);
        if (OWNER == 1 && lane == lstar) out_addr = lcring_base;
        if (OWNER == 2 && myslot >= 0) out_addr = wring_base + static_cast<uint32_t>(myslot * 64 * 4);
        const uint32_t half_even = static_cast<uint32_t>((j & 1) * kRows * 4), half_odd = static_cast<uint32_t>(((j + 1) & 1) * kRows * 4);
        if constexpr (OWNER == 1) {
            // entry 0 of this block's half: the last row of the block before -- still pending from the body before, or -1e9 in
            // every label column if that block was skipped (or there is none)
            *reinterpret_cast<float*>(smem + out_addr + half_even) = own_have ? own_pend : kProbMax;
            own_have = false;
        } else if constexpr (OWNER == 2) {
            if (!ring_has_row0)   // the block before was skipped (or there is none): its last row is -1e9 in every label column
                *reinterpret_cast<float*>(smem + out_addr + half_even) = kProbMax;
            ring_has_row0 = true;
        }
        float own_val = 0.0f;   // OWNER 1: last-column scores of the block before, on their way from the ring to HBM
        uint32_t* bp = bits + sd.bits_off + (int64_t)j * Cpad + cbase;   // trace words of block j, lane 0 (then + Cpad): wave-uniform

        // software pipeline: operands of row i+PF are requested while row i is computed
        constexpr int PF = CTCFA_PF;
        float2 emq[PF][K];
        float eq[PF][K];   // E_ALONE: e alone ...
        float2 xq[PF];     // ... and the row's (blank posterior, start-column stay step)
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            if constexpr (E_ALONE) xq[d] = *reinterpret_cast<const float2*>(smem + xaddr + d * ROW_BYTES);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (E_ALONE) eq[d][k] = *reinterpret_cast<const float*>(smem + gaddr[k] + d * ROW_BYTES);
                else emq[d][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + d * ROW_BYTES);
            }
        }
        int cnt_seen = 0;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int jj = i / kRows, ii = i % kRows, q = i / kHaloRows;   // block of the body, row of the block, group of the body
#if defined(CTCFA_STAMP) && CTCFA_STAMP == 3   // where inside a block the cycles go: a stamp every 8 rows of block 40
            if (ii % 8 == 0 && (j + jj == CTCFA_STAMP_BLOCK || j + jj == CTCFA_STAMP_BLOCK + 1) && lane == 0 && blockIdx.x < 64)
                (reinterpret_cast<unsigned long long*>(lastcol) + (1 + j + jj - CTCFA_STAMP_BLOCK) * 64 * 16 * 8 + (blockIdx.x * 16 + w) * 8)[ii / 8] = __builtin_amdgcn_s_memtime();
#endif
            if (i % kHaloRows == 0 && !(CTCFA_ABL & 1)) {   // group start: the neighbour's columns replace what went wrong in my halo
#pragma unroll
                for (int k = 0; k < K; ++k) prev[k] = is_halo ? hx[k] : prev[k];
            }
            float2 em[K];
            if constexpr (E_ALONE) {
                const float2 xr = xq[i % PF];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float e = eq[i % PF][k];
                    em[k] = make_float2(e, max3f((START && startlike[k]) ? xr.y : xr.x, e, kProbMax));
                }
                if (i + PF < ROWS) {
                    xq[i % PF] = *reinterpret_cast<const float2*>(smem + xaddr + (i + PF) * ROW_BYTES);
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        eq[i % PF][k] = *reinterpret_cast<const float*>(smem + gaddr[k] + (i + PF) * ROW_BYTES);
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) em[k] = emq[i % PF][k];
                if (i + PF < ROWS) {
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        emq[i % PF][k] = *reinterpret_cast<const float2*>(smem + gaddr[k] + (i + PF) * ROW_BYTES);
                }
            }
            if constexpr (OWNER == 1 && CTCFA_OWNER_DEFER && !(CTCFA_ABL & 4) && !(CTCFA_ABL & 256)) {   // (tuning, 256: the owner's stores alone left out)
                if (i > 0) {   // the score of row i - 1 (a register nobody waits for any more) goes to the ring: entry ii' + 1 of its block's half, the block's last row to entry 0 of the other
                    const int pj = (i - 1) / kRows, pi = (i - 1) % kRows;
                    const uint32_t hc = (pj & 1) ? half_odd : half_even, hn = (pj & 1) ? half_even : half_odd;
                    *reinterpret_cast<float*>(smem + out_addr + (pi + 1 < kRows ? hc + static_cast<uint32_t>((pi + 1) * 4) : hn)) = own_pend;
                }
                // the block before this one: from the ring half nobody writes to now, to HBM -- read here, stored four rows on
                if (ii == 8 && (jj > 0 || own_copy))
                    own_val = *reinterpret_cast<const float*>(smem + lcring_base + (((j + jj - 1) & 1) * kRows + (lane & 31)) * 4);
                if (ii == 12 && (jj > 0 || own_copy)) {
                    const int t = (j + jj - 1) * kRows + lane;
                    if (lane < kRows && t >= 1 && t < T) seg_lastcol[t] = own_val;
                }
            }
            // lane 0 has no left neighbour: it is a halo lane (its first column goes wrong at once, by
            // design) or left padding (e = -inf: the sum loses whatever comes in)
            const float leftv = dpp_wave_shr1_zero(prev[K - 1]);
#pragma unroll
            for (int k = K - 1; k >= 0; --k) {
                const float pl = (k == 0) ? leftv : prev[k > 0 ? k - 1 : 0];
                const float pk = prev[k];
                const float a = pl + em[k].x;
                const float b = pk + em[k].y;
                const float nw = max3f(a, b, kProbMax);
                if constexpr (!CK) {
                    const float rsw = em[k].x - (nw - pl);
                    const float rst = em[k].y - (nw - pk);
                    // sign bit of (|rsw| - |rst|) == (|rst| > |rsw|): SWITCH; ties -> STAY
                    const float d = __builtin_fabsf(rsw) - __builtin_fabsf(rst);
                    dec[k] = __builtin_amdgcn_alignbit(dec[k], __float_as_uint(d), 31);
                }
                prev[k] = nw;
            }
            if constexpr (OWNER == 1) {
                own_pend = prev[K - 1];
#if !CTCFA_OWNER_DEFER
                if (!(CTCFA_ABL & 4) && !(CTCFA_ABL & 256)) {   // (rounds 1-3: the score goes to the ring at once)
                    const uint32_t hc = (jj & 1) ? half_odd : half_even, hn = (jj & 1) ? half_even : half_odd;
                    *reinterpret_cast<float*>(smem + out_addr + (ii + 1 < kRows ? hc + static_cast<uint32_t>((ii + 1) * 4) : hn)) = own_pend;
                }
#endif
            }
            if constexpr (OWNER == 2) {
                float pv = prev[K - 1];
#pragma unroll
                for (int k = 0; k < K - 1; ++k) pv = (ksel == k) ? prev[k] : pv;
                const uint32_t hc = (jj & 1) ? half_odd : half_even, hn = (jj & 1) ? half_even : half_odd;
                *reinterpret_cast<float*>(smem + out_addr + (ii + 1 < kRows ? hc + static_cast<uint32_t>((ii + 1) * 4) : hn)) = pv;
            }
            // Pin this row's decisions here (empty asm = opaque use, no instruction): dec[] is
            // consumed at the end of the block, and LLVM otherwise sinks the residual math of
            // all 32 rows down to that store, keeping every operand alive (hundreds of spills).
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if constexpr (CK) asm volatile("" : "+v"(prev[k]));  // (keeps the rows apart for the scheduler)
                else asm volatile("" : "+v"(dec[k]));
            }
            if constexpr (!kLean) {
                // rounds 2-3's hand-over: the counter is read 7 rows before the group's end, looked at 4 rows before it (a spin
                // if the neighbour is not there yet), and only then the exchange row is read
                const int g = g0 + q;
                if (i % kHaloRows == kHaloRows - 1 - kPollLead - 3) {
                    if (w > 0 && !(CTCFA_ABL & 1)) cnt_seen = flags[w - 1];
                    if (i == ROWS - 1 - kPollLead - 3) {   // (for the next body: no wait at its start)
                        peek_sa = flags[16];
                        peek_sb = flags[17];
                    }
                }
                if (i % kHaloRows == kHaloRows - 1 - kPollLead && !(CTCFA_ABL & 1)) {
                    if (w > 0) {   // my neighbour's columns at the end of this group, for my next one
                        CTCFA_TRACE(w, 256 + g);
                        if (__builtin_expect(__builtin_amdgcn_readfirstlane(cnt_seen) < g + 1, 0)) {   // (normally it is 4+ rows ahead)
                            CTCFA_STAMP_BEGIN();
                            int f, spins = 0;
                            do {
#if CTCFA_NBR_SLEEP > 0
                                __builtin_amdgcn_s_sleep(CTCFA_NBR_SLEEP);
#endif
                                f = __builtin_amdgcn_readfirstlane(flags[w - 1]);
                                if (++spins > kSpinCap) { CTCFA_SPIN_DIAG("tile-nbr", w, g, f); break; }
                            } while (f < g + 1);
                            CTCFA_STAMP_END(st_nbr, st_nbr_n);
                            CTCFA_TRACE(w, 512 + g);
                        }
                        asm volatile("" ::: "memory");
                        const float* xr = reinterpret_cast<const float*>(smem + xin_cur + static_cast<uint32_t>(q * XW * 4));
#pragma unroll
                        for (int k = 0; k < K; ++k) hx[k] = xr[k];
                    }
                }
            }
            if (kLean && i % kHaloRows == kHaloRows - 1 - kPollLead) {
                // my neighbour's counter, then its columns at the end of this group (for my next one): read now, looked at at
                // the group's end -- no LDS round trip in the way
                CTCFA_TRACE(w, 256 + g0 + q);
                if (!(CTCFA_ABL & 1)) {
                cnt_seen = *cnt_in;
                asm volatile("" ::: "memory");
                const float* xr = reinterpret_cast<const float*>(smem + xin_cur + static_cast<uint32_t>(q * XW * 4));
#pragma unroll
                for (int k = 0; k < K; ++k) hx[k] = xr[k];
                }
                if (i == ROWS - 1 - kPollLead) {   // (for the next body: no wait at its start)
                    peek_sa = flags[16];
                    peek_sb = flags[17];
                }
            }
            if (i % kHaloRows == kHaloRows - 1) {   // group end: my last columns for the tile to my right, then the counter
                const int g = g0 + q;
                CTCFA_TRACE(w, g);
                if (CTCFA_ABL & 1) {
                    if (ii == kRows - 1 && lane == 63) flags[w] = g + 1;   // (the producer still wants to know)
                } else
                if constexpr (kAllLanesPublish) {
                    // every lane stores (the lanes that publish nothing: into the sink) -- no branch around the data, no exec juggling.
                    // Narrow tiles only: with more columns per lane 64 lanes' worth of LDS writes cost more than the branch
                    // (K = 3, 4096 segments: +9 %).
                    float* xw = reinterpret_cast<float*>(smem + xw_cur + static_cast<uint32_t>(q * XW * 4));
#pragma unroll
                    for (int k = 0; k < K; ++k) xw[k] = prev[k];
                    asm volatile("" ::: "memory");
                    *((lds_vint*)(smem + cnt_out_addr)) = g + 1;
                } else if (publishes) {
                    float* xw = reinterpret_cast<float*>(smem + xw_cur + static_cast<uint32_t>(q * XW * 4));
#pragma unroll
                    for (int k = 0; k < K; ++k) xw[k] = prev[k];
                    asm volatile("" ::: "memory");
                    if (lane == 63) flags[w] = g + 1;
                }
                if (kLean && !(CTCFA_ABL & 1) && __builtin_expect(__builtin_amdgcn_readfirstlane(cnt_seen) < g + 1, 0)) {   // (normally the neighbour is 4+ rows ahead)
                    CTCFA_STAMP_BEGIN();
                    int f, spins = 0;
                    do {
#if CTCFA_NBR_SLEEP > 0
                        __builtin_amdgcn_s_sleep(CTCFA_NBR_SLEEP);
#endif
                        f = __builtin_amdgcn_readfirstlane(*cnt_in);
                        if (++spins > kSpinCap) { CTCFA_SPIN_DIAG("tile-nbr", w, g, f); break; }
                    } while (f < g + 1);
                    CTCFA_STAMP_END(st_nbr, st_nbr_n);
                    CTCFA_TRACE(w, 512 + g);
                    asm volatile("" ::: "memory");
                    const float* xr = reinterpret_cast<const float*>(smem + xin_cur + static_cast<uint32_t>(q * XW * 4));
#pragma unroll
                    for (int k = 0; k < K; ++k) hx[k] = xr[k];
                    // (used here, so that the wait for it sits on this path: where the two paths meet again the compiler would
                    // otherwise wait for EVERY outstanding LDS read -- the prefetch queue drained at each group's start)
#pragma unroll
                    for (int k = 0; k < K; ++k) asm volatile("" : "+v"(hx[k]));
                }
                if (ii == kRows - 1) {   // end of block j + jj
                    // trace words of this block (fire and forget: this wave never waits on vmcnt); halo lanes hold
                    // copies that have gone wrong by now
#if defined(CTCFA_STAMP) && CTCFA_STAMP == 3
                    if ((j + jj == CTCFA_STAMP_BLOCK || j + jj == CTCFA_STAMP_BLOCK + 1) && lane == 0 && blockIdx.x < 64)
                        (reinterpret_cast<unsigned long long*>(lastcol) + (1 + j + jj - CTCFA_STAMP_BLOCK) * 64 * 16 * 8 + (blockIdx.x * 16 + w) * 8)[4] = __builtin_amdgcn_s_memtime();
#endif
                    if (lane >= HL && !(CTCFA_ABL & 4)) {
#pragma unroll
                        for (int k = 0; k < K; ++k) CTCFA_TRACE_STORE(bp + lane * K + k, CK ? __float_as_uint(prev[k]) : dec[k]);
                    }
                    bp += Cpad;
#if !CTCFA_OWNER_DEFER
                    if constexpr (OWNER == 1) {  // (rounds 1-3) last-column scores for the end-cell argmax: rows 32(j+jj) .. +31 are complete
                        const int t = (j + jj) * kRows + lane;
                        if (lane < kRows && t >= 1 && t < T)
                            seg_lastcol[t] = *reinterpret_cast<const float*>(smem + lcring_base + (((j + jj) & 1) * kRows + lane) * 4);
                    }
#endif
                    if constexpr (OWNER == 2) watch_out(j + jj, true);
                }
            }
        }
        if constexpr (OWNER == 1) {
            own_have = true;    // (own_pend: the body's last row)
            own_copy = CTCFA_OWNER_DEFER != 0;    // the body's last block is complete in the ring but for its last row, which belongs to the next block's entry 0
        }
    };

    // (only tile 0 holds the start column: the other tiles' code carries no select for it -- one vector instruction per
    // cell less on the tiles that are waited for)
    const bool tile_has_start = E_ALONE && cbase - shift <= 0;
    auto block = [&](int j, auto owner_tag, auto nb_tag) {
        if constexpr (E_ALONE) {
            if (tile_has_start) block_impl(j, owner_tag, std::true_type{}, nb_tag);
            else block_impl(j, owner_tag, std::false_type{}, nb_tag);
        } else {
            block_impl(j, owner_tag, std::false_type{}, nb_tag);
        }
    };
    // two-block bodies: the (e, m) pair kernels (their loop overhead is what a narrow tile notices); a ring with an even
    // number of slots, a body that starts in an even slot, both blocks staged already, both inside the tile's live range
    constexpr bool kPairs = (VP == 32 || (CTCFA_LEAN_ALL_PITCHES && VP <= 64)) && CTCFA_BODY_BLOCKS >= 2;
    const bool ring_even = (NS & 1) == 0;
    auto body = [&](int j, auto nb_tag) {
        if constexpr (K <= kWatchMaxK) {
            if (watch_tile) {
                block(j, std::integral_constant<int, 2>{}, nb_tag);
                return;
            }
        }
        if (w == wstar && wn == 0) block(j, std::integral_constant<int, 1>{}, nb_tag);
        else block(j, std::integral_constant<int, 0>{}, nb_tag);
    };
    for (int j = 0; j <= jlast;) {
        staged_seen = __builtin_amdgcn_readfirstlane(peek_sa < peek_sb ? peek_sa : peek_sb);
        if (CTCFA_ABL & 2) staged_seen = j + 2;
        if (__builtin_expect(staged_seen <= j, 0)) {   // emissions of block j (normally seen staged while block j-1 was computed)
            CTCFA_STAMP_BEGIN();
            for (int spins = 0;; ++spins) {
                staged_seen = staged_now();
                if (staged_seen > j) break;
                __builtin_amdgcn_s_sleep(1);
                if (spins > kSpinCap) { CTCFA_SPIN_DIAG("tile-staged", w, j, staged_seen); break; }
            }
            CTCFA_STAMP_END(st_staged, st_staged_n);
        }
        asm volatile("" ::: "memory");
        if (j < jfirst) {
            if (__builtin_amdgcn_readfirstlane(*posflag) == 0) {   // still provably -1e9 everywhere in this block
                if (wn > 0) {
                    watch_out(j, false);
                } else if (w == wstar) {
                    const int t = j * kRows + lane;
                    if (lane < kRows && t >= 1 && t < T) seg_lastcol[t] = kProbMax;   // (own_copy / own_have stay false: nothing computed yet)
                }
                if constexpr (CK) {  // the table row this block would have ended in
                    if (lane >= HL) {
                        uint32_t* bp = bits + sd.bits_off + (int64_t)j * Cpad + cbase + lane * K;
#pragma unroll
                        for (int k = 0; k < K; ++k) CTCFA_TRACE_STORE(bp + k, __float_as_uint(kProbMax));
                    }
                }
                if (lane == 63) flags[w] = kGroups * (j + 1);   // (its exchange rows are still the initial -1e9)
                ++j;
                jslot = (jslot + 1 == NS) ? 0 : jslot + 1;
                continue;
            }
            jfirst = 0;            // emissions are not log-probabilities: compute everything from here on
        }
        int nb = 1;
        if constexpr (kPairs) {
            if (ring_even && (jslot & 1) == 0 && j + 1 <= jlast && staged_seen > j + 1) nb = 2;
        }
        if constexpr (kPairs) {
            if (nb == 2) body(j, std::integral_constant<int, 2>{});
            else body(j, std::integral_constant<int, 1>{});
        } else {
            body(j, std::integral_constant<int, 1>{});
        }
        j += nb;
        jslot += nb;
        if (jslot >= NS) jslot -= NS;
#ifdef CTCFA_STAMP   // when the first two computed blocks ended: what a cold instruction cache costs a launch
        if (st_first[0] == 0) st_first[0] = __builtin_amdgcn_s_memtime() - st_t0;
        else if (st_first[1] == 0) st_first[1] = __builtin_amdgcn_s_memtime() - st_t0;
#endif
    }
    if (own_copy) {   // the owner's last block: from the ring to HBM
        const int t = jlast * kRows + lane;
        if (lane < kRows && t >= 1 && t < T)
            seg_lastcol[t] = *reinterpret_cast<const float*>(smem + lcring_base + ((jlast & 1) * kRows + lane) * 4);
    }
    if (lane == 63) flags[w] = kBigCount;   // done (end of the segment or dead zone): nobody waits for this tile again
    CTCFA_SPIN_REPORT();
#ifdef CTCFA_STAMP
    if (lane == 0 && blockIdx.x < 64) {
        unsigned long long* o = reinterpret_cast<unsigned long long*>(lastcol) + (blockIdx.x * 16 + w) * 8;
        o[0] = __builtin_amdgcn_s_memtime() - st_t0; o[1] = st_nbr; o[2] = st_staged; o[3] = st_nbr_n; o[4] = st_staged_n;
        o[5] = (st_first[0] << 32) | (st_first[1] & 0xffffffffull); o[6] = jlast; o[7] = st_t0;
    }
    return;
#endif
    // row 32*nblk (present when (T-1) % 32 == 0): what the watched column holds after the last block
    float pv_end = prev[K - 1];
#pragma unroll
    for (int k = 0; k < K - 1; ++k) pv_end = (ksel == k) ? prev[k] : pv_end;
    if (wn > 0) {
        if (myslot >= 0 && nblk * kRows < T)
            lastcol[*reinterpret_cast<const int64_t*>(smem + wtab_base + myslot * 8) + nblk * kRows] = pv_end;
    } else if (w == wstar && lane == lstar && nblk * kRows < T) {
        seg_lastcol[nblk * kRows] = pv_end;
    }
}

#ifndef CTCFA_FILL_KERNEL_ONLY   // (ctcfa_fill_pitches.hip: translation units that hold instantiations of fill_kernel and nothing else)

// ---------------------------------------------------------------------------------------
// Fill kernel for wide vocabularies (V > 128: sub-word CTC models).  A vocabulary row no longer
// fits an LDS ring, so nothing is staged: every lane owns ONE label column (K = 1) and gathers
// its own emission lpz[t, g_c] straight from HBM/L2, one 32-row block ahead of its use (two
// register sets, alternating blocks); the blank entry of the block's rows rides in lanes 0..31
// and reaches the row loop through v_readlane.  No producer wave, hence no posflag: only the
// input-independent dead-zone skip is kept.  Column 0 of the trellis is not a DP column here:
// with preamble_transition_cost_zero it is identically 0 and simply pre-fills the boundary ring
// of stage 0 (SegDesc.shift == -1: padded column pc holds label column pc + 1).
// Same recurrence, decisions, boundary exchange and decision-word layout as fill_kernel.
// ---------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
fill_gather_kernel(const SegDesc* __restrict__ segs, const float* __restrict__ lpz,
                   const int32_t* __restrict__ labels, uint32_t* __restrict__ bits,
                   float* __restrict__ lastcol, int V, int blank, int Cpad) {
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int W = blockDim.x >> 6;
    const SegDesc sd = segs[blockIdx.x];
    if (sd.prestatus != 0) return;  // uniform
    const int T = sd.T, C = sd.C;
    const unsigned char* lpz_bytes = reinterpret_cast<const unsigned char*>(lpz + sd.lpz_off);
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off;
    float* __restrict__ seg_lastcol = lastcol + sd.frm_off;
    const uint32_t bnd_base = 0u;
    const uint32_t lcring_base = static_cast<uint32_t>((W + 1) * kBndPitch * 4);
    const uint32_t sink_base = lcring_base + 64 * 4;
    float* bnd = reinterpret_cast<float*>(smem);
    const int nblk = (T - 1 + kRows - 1) / kRows;
    const int nsteps = nblk + W - 1;
    // ring 0 = trellis column 0 (table[t,0] == 0 with the preamble flag), rings 1.. = -1e9
    for (int i = tid; i < (W + 1) * kBndPitch; i += blockDim.x) bnd[i] = (i < kBndPitch) ? 0.0f : kProbMax;

    const int pc = w * 64 + lane;
    const int c = pc + 1;  // shift == -1
    const uint32_t goff = static_cast<uint32_t>((c < C) ? seg_lab[c] : blank) * 4u;
    float prev = kProbMax;  // table[0, c >= 1]
    uint32_t dec = 0u;
    const int wstar = sd.owner_stage;
    const int lstar = sd.owner_lane;
    float4 pub4 = make_float4(kProbMax, kProbMax, kProbMax, kProbMax);
    lds_barrier();
    if (w >= (W + 1) / 2) __builtin_amdgcn_s_setprio(2);
    else __builtin_amdgcn_s_setprio(1);

    int jlast = nblk - 1;  // dead zone, as in fill_kernel
    {
        const int cmax = w * 64 + 64;  // label column of lane 63
        if (cmax < C - 1) {
            const int tdead = T - C + cmax;
            jlast = tdead >= 1 ? (tdead - 1) / kRows : -1;
            if (jlast > nblk - 1) jlast = nblk - 1;
        }
    }
    if (w > wstar) jlast = -1;

    auto gload = [&](int jb, float (&e)[kRows], float& lbv) {
        const int t0 = jb * kRows + 1;
#pragma unroll
        for (int r = 0; r < kRows; ++r) {
            int t = t0 + r;
            t = t < T ? t : T - 1;  // rows past the end: re-read, results unused
            e[r] = *reinterpret_cast<const float*>(lpz_bytes + static_cast<uint64_t>(t) * static_cast<uint64_t>(V) * 4u + goff);
        }
        int tb = t0 + (lane & 31);
        tb = tb < T ? tb : T - 1;
        lbv = *reinterpret_cast<const float*>(lpz_bytes + static_cast<uint64_t>(tb) * static_cast<uint64_t>(V) * 4u +
                                              static_cast<uint32_t>(blank) * 4u);
    };
    auto block = [&](int j, const float (&e)[kRows], float lbv) {
        const int q = j & 3;
        const uint32_t in_addr = bnd_base + static_cast<uint32_t>((w * kBndPitch + q * kRows) * 4);
        uint32_t out_addr = sink_base + static_cast<uint32_t>(lane * 16);
        if (w == wstar) {
            if (lane == lstar) out_addr = lcring_base + static_cast<uint32_t>((j & 1) * kRows * 4);
        } else if (lane == 63 && w < wstar) {
            out_addr = bnd_base + static_cast<uint32_t>(((w + 1) * kBndPitch + q * kRows) * 4);
        }
        float4 lin4 = *reinterpret_cast<const float4*>(smem + in_addr);
        float4 lin4_next = lin4;
#pragma unroll
        for (int i = 0; i < kRows; ++i) {
            if (i % 4 == 0) {
                if (i > 0) lin4 = lin4_next;
                if (i + 4 < kRows) lin4_next = *reinterpret_cast<const float4*>(smem + in_addr + (i + 4) * 4);
            }
            const float lin = (i % 4 == 0) ? lin4.x : (i % 4 == 1) ? lin4.y : (i % 4 == 2) ? lin4.z : lin4.w;
            const float pl = dpp_wave_shr1(lin, prev);
            const float lb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(lbv), i));
            const float ee = e[i];
            const float m = max3f(lb, ee, kProbMax);
            const float a = pl + ee;
            const float b = prev + m;
            const float nw = max3f(a, b, kProbMax);
            const float rsw = ee - (nw - pl);
            const float rst = m - (nw - prev);
            const float d = __builtin_fabsf(rsw) - __builtin_fabsf(rst);
            dec = __builtin_amdgcn_alignbit(dec, __float_as_uint(d), 31);
            prev = nw;
            if ((i + 1) % 4 == 0) pub4.x = prev;
            else if ((i + 1) % 4 == 1) pub4.y = prev;
            else if ((i + 1) % 4 == 2) pub4.z = prev;
            else {
                pub4.w = prev;
                *reinterpret_cast<float4*>(smem + out_addr + (i - 2) * 4) = pub4;
            }
            asm volatile("" : "+v"(dec));
        }
        bits[sd.bits_off + (int64_t)j * Cpad + pc] = dec;
        if (w == wstar) {
            const int t = j * kRows + lane;
            if (lane < kRows && t >= 1 && t < T)
                seg_lastcol[t] = *reinterpret_cast<const float*>(smem + lcring_base + ((j & 1) * kRows + lane) * 4);
        }
        if (j == jlast && jlast < nblk - 1 && lane == 63 && w < wstar)
            bnd[(w + 1) * kBndPitch + ((j + 1) * kRows) % kBnd] = pub4.x;
    };

    for (int i = 0; i < w; ++i) lds_barrier();  // stage w starts w steps late
    float ea[kRows], eb[kRows];
    float la = 0.0f, lbb = 0.0f;
    if (0 <= jlast) gload(0, ea, la);
    int done = 0;
    for (int j = 0; j < nblk; j += 2) {
        if (j + 1 <= jlast) gload(j + 1, eb, lbb);
        if (j <= jlast) block(j, ea, la);
        lds_barrier();
        ++done;
        if (j + 1 >= nblk) break;
        if (j + 2 <= jlast) gload(j + 2, ea, la);
        if (j + 1 <= jlast) block(j + 1, eb, lbb);
        lds_barrier();
        ++done;
    }
    for (int i = w + done; i < nsteps; ++i) lds_barrier();
    if (w == wstar && lane == lstar && nblk * kRows < T) seg_lastcol[nblk * kRows] = pub4.x;
}

// ---------------------------------------------------------------------------------------
// Backtrack + per-frame outputs + utterance scoring.  grid = B, block = kBtThreads (4 waves).
//   phase 0  end cell: first maximum of the last column            (all threads)
//   phase A  the walk: one scalar step per run of STAYs            (wave 0; others wait)
//   phase B  char_probs / state / frame_of_label, lanes = frames   (all threads)
//   phase C  determine_utterance_segments, utterances over waves, lanes = window starts
// dynamic LDS: rec[nblk] (entry column, switch mask per 32-row block) + char_probs copy [T]
// ---------------------------------------------------------------------------------------
constexpr int kBtThreads = 256;
constexpr uint32_t kBtFlagLowPriority = 0x100u;   // BtParams.flags: run below the fill tiles of the next batch (host decides)

struct BtParams {
    int V, blank, Cpad;
    uint32_t flags;
    int L;            // score_min_mean_over_L
    int rec_bytes;    // bytes reserved for rec[] (multiple of 16) in dynamic LDS
    int lab_bytes;    // checkpoint mode: bytes reserved for the label copy that follows rec[] (multiple of 16), else 0
    int fol_bytes;    // checkpoint mode: bytes reserved for the LDS copy of frame_of_label (multiple of 16)
    int scorers;      // checkpoint mode: waves of the workgroup (the last ones) that only score utterances
    int prio;         // checkpoint mode: s_setprio of the striders (0..3)
    int windows;      // checkpoint mode: windows a strider may recompute for one block while it waits for the entry column (1..3)
    double dur;       // index_duration
};

// (Checkpoint mode -- fill_kernel<.., CK = true>, V <= 64: the fill stores no decisions, only the table row every
// 32-row block ends in -- has its own backtrack kernel, stride_backtrack_kernel below.)

__device__ __forceinline__ float dpp_wave_shl1(float src) {
    // lane i <- src[lane i+1]; lane 63 <- 0 (bound_ctrl: the DPP folds into the consuming VALU op)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x130, 0xf, 0xf, true));
}

__device__ __forceinline__ double np_pairwise_sum_le128(const float* a, int n) {
    // NumPy's pairwise summation for n <= 128 (fp64 accumulate of fp32-exact values), LDS input
    if (n < 8) {
        double res = 0.0;
        for (int i = 0; i < n; ++i) res += (double)a[i];
        return res;
    }
    double r[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = (double)a[q];
    int i;
    for (i = 8; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q) r[q] += (double)a[i + q];
    }
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += (double)a[i];
    return res;
}

// NumPy's pairwise summation for any n: above 128 elements the vector is split at n/2 rounded down to a multiple
// of 8 and the halves are summed on their own (numpy/core/src/umath/loops_utils.h.src, pairwise_sum) -- the
// recursion as a loop over an explicit stack.  Only the rescoring kernel of score_min_mean_over_L > 128 uses it.
__device__ inline double np_pairwise_sum_any(const float* a, int n) {
    constexpr int kDepth = 24;
    const float* fa[kDepth];
    int fn[kDepth];
    int fs[kDepth];
    double fv[kDepth];
    int sp = 1;
    fa[0] = a;
    fn[0] = n;
    fs[0] = 0;
    double ret = 0.0;
    while (sp > 0) {
        const int k = sp - 1;
        if (fs[k] == 0) {
            if (fn[k] <= 128 || sp >= kDepth) {
                ret = np_pairwise_sum_le128(fa[k], fn[k]);
                --sp;
            } else {
                int n2 = fn[k] / 2;
                n2 -= n2 % 8;
                fs[k] = 1;
                fa[sp] = fa[k];
                fn[sp] = n2;
                fs[sp] = 0;
                ++sp;
            }
        } else if (fs[k] == 1) {
            int n2 = fn[k] / 2;
            n2 -= n2 % 8;
            fv[k] = ret;
            fs[k] = 2;
            fa[sp] = fa[k] + n2;
            fn[sp] = fn[k] - n2;
            fs[sp] = 0;
            ++sp;
        } else {
            ret = fv[k] + ret;
            --sp;
        }
    }
    return ret;
}

// determine_utterance_segments() of ctc-segmentation 1.7.1 for one segment: utterances over
// waves, lanes = sliding windows.  `fol` = frame of every label column (read with agent-scope
// loads: written by other waves of this workgroup), `cps` = the segment's char_probs in LDS.
struct NoTick {
    __device__ __forceinline__ void operator()() const {}
};

template <int NTHREADS, class Tick = NoTick, bool ANY_L = false>
__device__ __forceinline__ void score_utterances(const SegDesc& sd, int L, double dur, const int32_t* ub,
                                                 const int32_t* fol, const float* cps, int T, int C, int U,
                                                 double* __restrict__ seg_start, double* __restrict__ seg_end,
                                                 double* __restrict__ seg_score, int tid = -1, Tick tick = Tick(),
                                                 int nwaves_rt = 0) {
    if (tid < 0) tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = NTHREADS > 0 ? NTHREADS / 64 : nwaves_rt;   // (NTHREADS == 0: the caller's wave count is a run-time value)
    auto tim = [&](int c) {
        if (c < 0) c += C;  // NumPy wrap (never taken for well-formed utt_begin)
        return (double)__hip_atomic_load(fol + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * dur;
    };
    const int n = L;
    for (int u = wave; u < U; u += NW) {
        const int b = ub[u], e = ub[u + 1];
        const double mid_b = (tim(b) + tim(b - 1)) / 2;
        const double sb = tim(b + 1) - 0.5;
        const double start = (mid_b > sb) ? mid_b : sb;  // max(timings[b+1]-0.5, middle)
        const double mid_e = (tim(e) + tim(e - 1)) / 2;
        const double ee = tim(e - 1) + 0.5;
        const double end = (mid_e < ee) ? mid_e : ee;  // min(timings[e-1]+0.5, middle)
        const long long start_t = (long long)rint(start / dur);
        const long long end_t = (long long)rint(end / dur);
        double min_avg;
        if (end_t <= start_t) {
            min_avg = -10000000000.0;
        } else if (end_t - start_t <= n) {
            long long lo = start_t < 0 ? 0 : start_t, hi = end_t > T ? T : end_t;
            if (lo > T) lo = T;
            if (hi < lo) hi = lo;
            min_avg = (ANY_L ? np_pairwise_sum_any(cps + lo, (int)(hi - lo)) : np_pairwise_sum_le128(cps + lo, (int)(hi - lo))) / (double)(hi - lo);
        } else {
            double local = 0.0;
            int it = 0;
            for (long long t0 = start_t + lane; t0 < end_t - n; t0 += 64) {
                if ((++it & 1) == 0) tick();
                long long lo = t0 < 0 ? 0 : t0, hi = (t0 + n > T) ? T : t0 + n;
                if (lo > T) lo = T;
                if (hi < lo) hi = lo;
                const double m = (ANY_L ? np_pairwise_sum_any(cps + lo, (int)(hi - lo)) : np_pairwise_sum_le128(cps + lo, (int)(hi - lo))) / (double)(hi - lo);
                if (m < local) local = m;
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const double o = __shfl_xor(local, off);
                if (o < local) local = o;
            }
            min_avg = local;
        }
        if (lane == 0) {
            seg_start[sd.utt_off + u] = start;
            seg_end[sd.utt_off + u] = end;
            seg_score[sd.utt_off + u] = min_avg;
        }
        tick();
    }
}

// One utterance of determine_utterance_segments() by one wave (lanes = sliding windows); `fol` and `cps` in LDS.
__device__ __forceinline__ void score_one_utterance(const SegDesc& sd, int L, double dur, const int32_t* ub, const int32_t* fol,
                                                    const float* cps, int T, int C, int u, double* __restrict__ seg_start,
                                                    double* __restrict__ seg_end, double* __restrict__ seg_score, int lane) {
    auto tim = [&](int c) {
        if (c < 0) c += C;  // NumPy wrap (never taken for well-formed utt_begin)
        return (double)fol[c] * dur;
    };
    const int n = L;
    const int b = ub[u], e = ub[u + 1];
    const double mid_b = (tim(b) + tim(b - 1)) / 2;
    const double sb = tim(b + 1) - 0.5;
    const double start = (mid_b > sb) ? mid_b : sb;  // max(timings[b+1]-0.5, middle)
    const double mid_e = (tim(e) + tim(e - 1)) / 2;
    const double ee = tim(e - 1) + 0.5;
    const double end = (mid_e < ee) ? mid_e : ee;  // min(timings[e-1]+0.5, middle)
    const long long start_t = (long long)rint(start / dur);
    const long long end_t = (long long)rint(end / dur);
    double min_avg;
    if (end_t <= start_t) {
        min_avg = -10000000000.0;
    } else if (end_t - start_t <= n) {
        long long lo = start_t < 0 ? 0 : start_t, hi = end_t > T ? T : end_t;
        if (lo > T) lo = T;
        if (hi < lo) hi = lo;
        min_avg = np_pairwise_sum_le128(cps + lo, (int)(hi - lo)) / (double)(hi - lo);
    } else {
        double local = 0.0;
        for (long long t0 = start_t + lane; t0 < end_t - n; t0 += 64) {
            long long lo = t0 < 0 ? 0 : t0, hi = (t0 + n > T) ? T : t0 + n;
            if (lo > T) lo = T;
            if (hi < lo) hi = lo;
            const double m = np_pairwise_sum_le128(cps + lo, (int)(hi - lo)) / (double)(hi - lo);
            if (m < local) local = m;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(local, off);
            if (o < local) local = o;
        }
        min_avg = local;
    }
    if (lane == 0) {
        seg_start[sd.utt_off + u] = start;
        seg_end[sd.utt_off + u] = end;
        seg_score[sd.utt_off + u] = min_avg;
    }
}

// Everything the backtrack needs, as one argument: the stand-alone kernel below and the backtrack
// wave fused into the mixed-shape fill kernel run the same body.
struct BtArgs {
    const SegDesc* segs;
    const float* lpz;
    const int32_t* labels;
    const int32_t* utt_begin;
    const uint32_t* bits;
    const float* lastcol;
    const int32_t* fill_err;   // error word of the workspace: == fill_run if a wait of the fill that produced bits / lastcol gave up (NULL: none)
    int32_t fill_run;          // the number of this run
    BtParams p;
    int32_t* frame_of_label;
    float* char_prob;
    int32_t* state;
    double* seg_start;
    double* seg_end;
    double* seg_score;
    int32_t* t_end_out;
    int32_t* status_out;
    // Small host-buffer calls (stride_backtrack_kernel the last kernel of the call, results written straight into the pinned
    // result block): the workgroup that finishes last writes `done_value` to `done_word` (pinned, host-coherent), which the
    // host polls instead of waiting for the stream -- the runtime's completion signal costs a call ~3 us more.  NULL: off.
    uint32_t* done_count;   // device counter, never reset: this launch's workgroups take it to done_target
    uint32_t done_target;
    uint32_t* done_word;
    uint32_t done_value;
};

// NT cooperating threads (256: a workgroup of its own, 64: one wave inside a fill workgroup);
// `sync` separates the phases, `tick` is called between slices of work (the fused wave has to
// take part in the fill workgroup's barriers while it lives).
template <int NT, class Sync, class Tick>
__device__ __forceinline__ void backtrack_body(const BtArgs& a, const SegDesc& sd, int tid, unsigned char* smem,
                                               float* red_v, int* red_t, int* sh_misc, Sync sync, Tick tick) {
    const BtParams& p = a.p;
    const float* __restrict__ lpz = a.lpz;
    const int32_t* __restrict__ labels = a.labels;
    const int32_t* __restrict__ utt_begin = a.utt_begin;
    const uint32_t* __restrict__ bits = a.bits;
    const float* __restrict__ lastcol = a.lastcol;
    int32_t* __restrict__ frame_of_label = a.frame_of_label;
    float* __restrict__ char_prob = a.char_prob;
    int32_t* __restrict__ state = a.state;
    double* __restrict__ seg_start = a.seg_start;
    double* __restrict__ seg_end = a.seg_end;
    double* __restrict__ seg_score = a.seg_score;
    int32_t* __restrict__ t_end_out = a.t_end_out;
    int32_t* __restrict__ status_out = a.status_out;
    constexpr int kThreads = NT;
    int2* rec = reinterpret_cast<int2*>(smem);                   // per block: (entry column, switch mask)
    float* cps = reinterpret_cast<float*>(smem + p.rec_bytes);   // char_probs of this segment
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = kThreads / 64;
    const int T = sd.T, C = sd.C, U = sd.U, shift = sd.shift, V = p.V;
    const float* __restrict__ seg_lpz = lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off;
    int32_t* fol = frame_of_label + sd.lab_off;
    float* cp = char_prob + sd.frm_off;
    int32_t* st = state ? state + sd.frm_off : nullptr;
    const bool want_seg = (utt_begin != nullptr) && (seg_score != nullptr) && U > 0;

    auto fail = [&](int code) {
        for (int c = tid; c < C; c += kThreads) fol[c] = 0;
        for (int t = tid; t < T; t += kThreads) {
            cp[t] = 0.0f;
            if (st) st[t] = -2;
        }
        if (want_seg)
            for (int u = tid; u < U; u += kThreads) {
                seg_start[sd.utt_off + u] = 0.0;
                seg_end[sd.utt_off + u] = 0.0;
                seg_score[sd.utt_off + u] = 0.0;
            }
        if (tid == 0) {
            status_out[sd.seg_index] = code;
            t_end_out[sd.seg_index] = -1;
        }
    };
    if (sd.prestatus == kPreWindowed) return;  // windowed_kernel owns this segment
    if (sd.prestatus != 0) {
        fail(sd.prestatus);
        return;
    }
    if (a.fill_err && *a.fill_err == a.fill_run) {   // the fill gave up on a progress counter: its trace is not to be trusted
        fail(5);
        return;
    }

    // ---- phase 0: first maximum of the last column (lastMax/lastArgMax of the fill) --------
    {
        const float* lc = lastcol + sd.frm_off;
        float bv = -__builtin_inff();
        int bt = 0x7fffffff;
        int it0 = 0;
        for (int t = tid; t < T; t += kThreads) {
            if ((++it0 & 7) == 0) tick();
            const float v = (t == 0) ? kProbMax : lc[t];  // table[0, C-1] = -1e9
            if (bt == 0x7fffffff || v > bv) {             // ascending t per thread: strict '>' keeps the first
                bv = v;
                bt = t;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int ot = __shfl_xor(bt, off);
            if (ot != 0x7fffffff && (bt == 0x7fffffff || ov > bv || (ov == bv && ot < bt))) {
                bv = ov;
                bt = ot;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_t[wave] = bt;
        }
    }
    const int nblk = (T - 1 + kRows - 1) / kRows;
    for (int j = tid; j < nblk; j += kThreads) rec[j] = make_int2(-1, 0);
    for (int c = tid; c < C; c += kThreads) fol[c] = 0;
    sync();

    // ---- phase A (wave 0): the walk, one scalar step per run of STAYs ----------------------
    if (wave == 0) {
        float bv = red_v[0];
        int bt = red_t[0];
#pragma unroll
        for (int q = 1; q < NW; ++q) {
            const float ov = red_v[q];
            const int ot = red_t[q];
            if (ot != 0x7fffffff && (bt == 0x7fffffff || ov > bv || (ov == bv && ot < bt))) {
                bv = ov;
                bt = ot;
            }
        }
        int t_end = __builtin_amdgcn_readfirstlane(bt);
        if (p.flags & 4u) t_end = T - 1;
        int pc = C - 1 + shift;
        int bad = 0;
        if (t_end >= 1) {
            int j = (t_end - 1) >> 5;
            int b0 = 31 - ((t_end - 1) & 31);
            const uint32_t* seg_bits = bits + sd.bits_off;
            // Per 32-row block: lane i holds the decision word of column (pc - i) (the path can drop
            // at most 32 columns inside the block).  The words of a block are requested kDepth
            // blocks ahead, when its entry column is only known to lie within 32*kDepth columns
            // below the current one: three loads cover columns [base-191, base], and the lane index
            // is re-based with a shuffle once the block is reached.  (One block ahead was not
            // enough: a block is walked in ~0.2 us, a miss in the decision words costs ~1 us.)
            constexpr int kDepth = 4;
            auto fetch = [&](int jb, int base, int part) -> uint32_t {
                const int col = base - 64 * part - lane;
                // columns at or left of the start column (<= shift) always STAY: their bits read as 0
                return (jb >= 0 && col - shift > 0) ? seg_bits[(int64_t)jb * p.Cpad + col] : 0u;
            };
            uint32_t pf[kDepth][3];
            int pbase[kDepth];
#pragma unroll
            for (int u = 0; u < kDepth; ++u) {
                pbase[u] = pc;
#pragma unroll
                for (int q = 0; q < 3; ++q) pf[u][q] = fetch(j - u, pc, q);
            }
            while (j >= 0) {
#pragma unroll
                for (int u = 0; u < kDepth; ++u) {
                    if (j < 0) break;
                    const int cstart = pc;
                    // this block's words for columns pc - lane, out of the loads issued for slot u
                    const int src = (pbase[u] - pc) + lane;  // 0 .. 191
                    const uint32_t f0 = __shfl(pf[u][0], src & 63);
                    const uint32_t f1 = __shfl(pf[u][1], src & 63);
                    const uint32_t f2 = __shfl(pf[u][2], src & 63);
                    const uint32_t wl = (src < 64) ? f0 : (src < 128) ? f1 : f2;
                    // slot u is free again: request block j - kDepth relative to the column we are at
                    pbase[u] = pc;
#pragma unroll
                    for (int q = 0; q < 3; ++q) pf[u][q] = fetch(j - kDepth, pc, q);
                    // Walk the block switch by switch, not row by row: the word of the current column
                    // sits in lane pidx; its lowest set bit at or above the current row is the next
                    // SWITCH (every row in between STAYs), then the column index drops by one.
                    uint32_t S = 0;
                    int pidx = 0;  // columns dropped so far in this block
                    int bb = b0;
                    // (hand-scheduled: the compiler's version of this loop spends 16 scalar
                    // instructions and four branches per switch)
                    //   while (bb < 32) { w = readlane(wl, pidx) >> bb; if (!w) break;
                    //                     bb += ctz(w); S |= 1 << bb; ++pidx; ++bb; }
                    if (bb < kRows) {
                        uint32_t tmp;
                        asm volatile(
                            "1:\n\t"
                            "s_nop 3\n\t"                        // wl was just written by a VALU op: gfx940 needs a wait
                                                                 // state before v_readlane reads it (and the
                                                                 // hazard recogniser does not look inside asm)
                            "v_readlane_b32 %3, %4, %1\n\t"
                            "s_lshr_b32 %3, %3, %2\n\t"
                            "s_cmp_eq_u32 %3, 0\n\t"
                            "s_cbranch_scc1 2f\n\t"
                            "s_ff1_i32_b32 %3, %3\n\t"
                            "s_add_i32 %3, %3, %2\n\t"
                            "s_bitset1_b32 %0, %3\n\t"
                            "s_add_i32 %2, %3, 1\n\t"
                            "s_add_i32 %1, %1, 1\n\t"
                            "s_cmp_lt_i32 %2, 32\n\t"
                            "s_cbranch_scc1 1b\n\t"
                            "2:"
                            : "+s"(S), "+s"(pidx), "+s"(bb), "=&s"(tmp)
                            : "v"(wl)
                            : "scc");
                    }
                    pc -= pidx;
                    if (lane == 0) rec[j] = make_int2(cstart, (int)S);
                    --j;
                    b0 = 0;
                    if (u & 1) tick();
                }
            }
            if (pc - shift > 0) bad = 1;  // reached t == 0 in a label column: the package's IndexError
        } else {
            bad = (pc - shift > 0);
        }
        if (lane == 0) {
            sh_misc[0] = t_end;
            sh_misc[1] = bad;
        }
    }
    sync();
    const int t_end = sh_misc[0];
    if (sh_misc[1]) {
        fail(2);
        return;
    }

    // ---- phase B: per-frame outputs, lanes = frames ---------------------------------------
    int itb = 0;
    for (int t = tid; t < T; t += kThreads) {
        if ((++itb & 3) == 0) tick();
        float prob = 0.0f;
        int s_lab = -2;
        if (t >= 1 && t <= t_end) {
            const int j = (t - 1) >> 5;
            const int b = 31 - ((t - 1) & 31);
            const int2 r = rec[j];
            const uint32_t S = (uint32_t)r.y;
            const int pct = r.x - __builtin_popcount(S & ((1u << b) - 1u));
            const int sw = (S >> b) & 1u;
            const int c = pct - shift;
            const float lb = seg_lpz[(int64_t)t * V + p.blank];
            if (c <= 0) {
                prob = __builtin_fmaxf(lb, kMaxProb);
                s_lab = -1;
            } else {
                const int g = seg_lab[c];
                const float e = seg_lpz[(int64_t)t * V + g];
                const float mx = __builtin_fmaxf(e, kMaxProb);
                if (sw) {
                    prob = mx;
                    s_lab = g;
                    fol[c] = t;
                } else {
                    prob = (mx > lb) ? mx : lb;
                    s_lab = -1;
                }
            }
        }
        cp[t] = prob;
        cps[t] = prob;
        if (st) st[t] = s_lab;
    }
    if (tid == 0) {
        status_out[sd.seg_index] = 0;
        t_end_out[sd.seg_index] = t_end;
    }
    if (!want_seg) return;
    sync();

    // ---- phase C: determine_utterance_segments ------------------------------------------------
    score_utterances<kThreads>(sd, p.L, p.dur, utt_begin + sd.utt_off + sd.seg_index, fol, cps, T, C, U,
                               seg_start, seg_end, seg_score, tid, tick);
}

struct BlockSync {
    __device__ __forceinline__ void operator()() const {
        __threadfence_block();
        __syncthreads();
    }
};
struct WaveSync {  // one wave: program order + completed memory operations is all a phase change needs
    __device__ __forceinline__ void operator()() const {
        __threadfence_block();
        __builtin_amdgcn_wave_barrier();
    }
};

__global__ void __launch_bounds__(kBtThreads, 5)   // <= 96 VGPRs: two backtrack workgroups still fit a SIMD that carries five 64-register fill tiles
backtrack_kernel(BtArgs a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float red_v[kBtThreads / 64];
    __shared__ int red_t[kBtThreads / 64];
    __shared__ int sh_misc[4];  // [0] t_end, [1] bad
    const SegDesc sd = a.segs[blockIdx.x];
    backtrack_body<kBtThreads>(a, sd, (int)threadIdx.x, smem, red_v, red_t, sh_misc, BlockSync(), NoTick());
}


// ---------------------------------------------------------------------------------------
// Checkpoint-mode backtrack, round 3: "striders".
//
// The fill left table row 32j+32 of every block j (trace word [j][pc]).  The walk from the end cell
// to (0, 0) is one dependent chain through T rows; everything else a block needs -- its 64-column
// window of recurrence + residual test -- depends only on WHERE the path enters the block, and the
// path drops at most one column per row.  So NW waves (blockDim / 64, one 32-row block each,
// blocks dealt round-robin from the end cell downwards) recompute their block SPECULATIVELY while
// the walks of the NW-1 blocks above are still going on: the window is anchored on the latest
// entry column anybody has published, moved down by the columns the path is expected to drop on
// the way (its recent slope), lane l = column top - l.  Row r of a window is wrong in lanes >= 64-r
// (their left neighbours lie outside the wave); the path enters at lane x = top - entry and sits in
// lanes <= x + 31 - r, so any 0 <= x <= 31 is exact.  When the block's true entry column arrives
// (LDS word written by the wave that walked the block above) and x is outside that range, the
// block is recomputed from its exact entry column: speculation only ever costs time.
//   * decisions are the v_cmp masks themselves (one SGPR pair per row, lane = column): the walk is
//     4 scalar instructions per row (s_bitcmp1_b64 / s_addc_u32), no decision words, no LDS hop
//     between a "recompute" and a "walker" wave, no workgroup barrier in the loop;
//   * every wave stages the emission rows of its OWN next block: 32 V floats are contiguous in
//     lpz -- dwordx4 loads issued a whole turn ahead, written into the wave's private LDS slot (row
//     pitch P) after the block that used it; the start column's label (e = -inf) reads a column of
//     -inf kept beside the slots;
//   * per-frame outputs (char_probs, state, frame_of_label) of a block are written by the wave that
//     walked it, from the emission rows it still holds in LDS -- lpz is read once per block;
//   * end-cell argmax before, utterance scoring after (all waves), as in backtrack_kernel.
// dynamic LDS: rec[nblk] | labels (bytes) | -inf column [32][P] | NW slots [32][P] | char_probs [T]
// ---------------------------------------------------------------------------------------
constexpr int kSbMaxWaves = 8;
constexpr int kSbSentinel = (int)0x80000000;   // rec[j].x: entry column of block j not known yet
#ifndef CTCFA_SB_MARGIN
#define CTCFA_SB_MARGIN 15
#endif
#ifndef CTCFA_SB_RING
#define CTCFA_SB_RING 8    // emission rows a strider holds in registers ahead of the row it computes
#endif

struct __attribute__((packed, aligned(4))) F4U { float x, y, z, w; };   // four consecutive floats, dword aligned

template <class F, int... I>
__device__ __forceinline__ void for_each_row(F&& f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);   // rows in order, the row number a compile-time constant
}

// NARROW (a narrowed plan, P = 32; see narrow_build): a staged row holds the 32 vocabulary entries of the segment's ring
// (entry 0 the blank) instead of the whole vocabulary row -- gathered from lpz, four entries of four rows per lane -- and
// the label bytes are ring entries: the vocabulary itself may have up to 256 entries.  The ring -> vocabulary table (4 P
// bytes; P = 64: the 64-entry ring of texts with 32 .. 62 labels) lies between frame_of_label and the -inf column.
template <int P, bool NARROW>
__device__ __forceinline__ void stride_backtrack_body(const BtArgs& a);

template <int P, bool NARROW = false>   // P: LDS row pitch of a staged emission block (the vocabulary rounded up: 32 / 40 / 48 / 56 / 64)
__global__ void __launch_bounds__(64 * kSbMaxWaves, 5)   // <= 96 VGPRs: room beside the 64-register fill tiles of the next batch
stride_backtrack_kernel(BtArgs a) {
    stride_backtrack_body<P, NARROW>(a);
    if (a.done_word) {   // (uniform) every way out of the body ends here: the call's completion word
        __threadfence_system();   // this thread's results are in the host's memory before ...
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t before = __hip_atomic_fetch_add(a.done_count, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1u == a.done_target)   // ... the last workgroup says so
                __hip_atomic_store(a.done_word, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

template <int P, bool NARROW>
__device__ __forceinline__ void stride_backtrack_body(const BtArgs& a) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ float red_v[kSbMaxWaves];
    __shared__ int red_t[kSbMaxWaves];
    __shared__ int sh_misc[4];   // [0] t_end, [1] bad, [2] a wait gave up, [3] next utterance to score
#ifdef CTCFA_BT_STAMP
    __shared__ uint32_t sh_pubtime[256];   // when block j's entry column was published (low word of s_memtime)
#endif
    const SegDesc sd = a.segs[blockIdx.x];
    const BtParams& p = a.p;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int NW = (blockDim.x >> 6) - p.scorers;   // striders: waves 0 .. NW-1 (the waves after them only score)
    const int nthreads = blockDim.x;
    const bool ferr = a.fill_err ? *a.fill_err == a.fill_run : false;   // (asked for first: nothing else waits behind it)
    const int T = sd.T, C = sd.C, U = sd.U, shift = sd.shift, V = p.V;
    const float* __restrict__ seg_lpz = a.lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = a.labels + sd.lab_off;
    const uint32_t* __restrict__ seg_bits = a.bits + sd.bits_off;
    const int rblank = NARROW ? 0 : p.blank;   // the blank's place in a staged row
    int32_t* fol = a.frame_of_label + sd.lab_off;
    float* cp = a.char_prob + sd.frm_off;
    int32_t* st = a.state ? a.state + sd.frm_off : nullptr;
    const bool want_seg = (a.utt_begin != nullptr) && (a.seg_score != nullptr) && U > 0;
#ifdef CTCFA_BT_STAMP   // tuning builds: where a strider's cycles go (tools/bt_stamps.py reads them from the `state` buffer)
    unsigned long long* stamp_out = st ? reinterpret_cast<unsigned long long*>(a.state + sd.frm_off) : nullptr;
    st = nullptr;
    uint32_t sk[16] = {0};   // (low words of s_memtime) [0] start [1] end cell known [2] chain done [3] all done [4] scored; [5..] per-turn sums
    uint32_t s_last = (uint32_t)__builtin_amdgcn_s_memtime();
    sk[0] = s_last;
#define SB_LAP(k) do { const uint32_t now_ = (uint32_t)__builtin_amdgcn_s_memtime(); sk[k] += now_ - s_last; s_last = now_; } while (0)
#define SB_MARK(k) do { sk[k] = (uint32_t)__builtin_amdgcn_s_memtime(); } while (0)
#define SB_COUNT(k) do { sk[k] += 1; } while (0)
#else
#define SB_LAP(k) do { } while (0)
#define SB_MARK(k) do { } while (0)
#define SB_COUNT(k) do { } while (0)
#endif
    constexpr int SLOT_BYTES = kRows * P * 4;
    lds_vint* rec = (lds_vint*)smem;                           // [2j] entry column of block j, [2j+1] its switch mask
    uint8_t* labs = smem + p.rec_bytes;
    int32_t* fol_lds = reinterpret_cast<int32_t*>(smem + p.rec_bytes + p.lab_bytes);   // frame_of_label, for the scoring
    lds_vint* nrw = (lds_vint*)(smem + p.rec_bytes + p.lab_bytes + p.fol_bytes);   // NARROW: ring entry -> vocabulary entry
    const uint32_t neg_base = (uint32_t)(p.rec_bytes + p.lab_bytes + p.fol_bytes + (NARROW ? P * 4 : 0));
    const uint32_t slot0 = neg_base + SLOT_BYTES;
    float* cps = reinterpret_cast<float*>(smem + slot0 + NW * SLOT_BYTES);   // char_probs of this segment (scoring)

    auto fail = [&](int code) {
        for (int c = tid; c < C; c += nthreads) fol[c] = 0;
        for (int t = tid; t < T; t += nthreads) {
            cp[t] = 0.0f;
            if (st) st[t] = -2;
        }
        if (want_seg)
            for (int u = tid; u < U; u += nthreads) {
                a.seg_start[sd.utt_off + u] = 0.0;
                a.seg_end[sd.utt_off + u] = 0.0;
                a.seg_score[sd.utt_off + u] = 0.0;
            }
        if (tid == 0) {
            a.status_out[sd.seg_index] = code;
            a.t_end_out[sd.seg_index] = -1;
        }
    };
    if (sd.prestatus == kPreWindowed) return;  // windowed_kernel owns this segment
    if (sd.prestatus != 0) {
        fail(sd.prestatus);
        return;
    }
    if (ferr) {   // the fill gave up on a progress counter: its trace is not to be trusted
        fail(5);
        return;
    }

    // ---- end cell: first maximum of the last column (lastMax / lastArgMax of the package's fill) ----
    {
        const float* lc = a.lastcol + sd.frm_off;
        float bv = -__builtin_inff();
        int bt = 0x7fffffff;
        for (int t = tid; t < T; t += nthreads) {
            const float v = (t == 0) ? kProbMax : lc[t];  // table[0, C-1] = -1e9
            if (bt == 0x7fffffff || v > bv) {             // ascending t per thread: strict '>' keeps the first
                bv = v;
                bt = t;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int ot = __shfl_xor(bt, off);
            if (ot != 0x7fffffff && (bt == 0x7fffffff || ov > bv || (ov == bv && ot < bt))) {
                bv = ov;
                bt = ot;
            }
        }
        if (lane == 0) {
            red_v[wave] = bv;
            red_t[wave] = bt;
        }
    }
    const int nblk = (T - 1 + kRows - 1) / kRows;
    for (int j = tid; j < nblk; j += nthreads) {
        rec[2 * j] = kSbSentinel;
        rec[2 * j + 1] = 0;
    }
    bool too_many = false;
    if (NARROW) {   // every wave derives the segment's table by itself, in the slot of the first strider (nothing is staged yet)
        lds_vuint* scr = (lds_vuint*)(smem + slot0 + (uint32_t)(wave * kNarrowScratchBytes));
        static_assert(!NARROW || P == 32 || P == 64, "a narrowed plan's ring has 32 or 64 entries");
        narrow_build<P == 64 ? 64 : 32>(scr, seg_lab, C, V, p.blank, lane);
        too_many = narrow_count(scr) > narrow_max_labels(P);   // (the same answer in every wave)
        for (int c = tid; c < C; c += nthreads) labs[c] = (uint8_t)(c > 0 ? narrow_rank(scr, seg_lab[c], p.blank) & (P - 1) : 0);
        if (wave == 0 && lane < P) nrw[lane] = (int)scr[16 + lane];
    }
    for (int c = tid; c < C; c += nthreads) {
        fol[c] = 0;
        fol_lds[c] = 0;
        if (!NARROW) labs[c] = (uint8_t)seg_lab[c];  // [0] = -1 is never looked up
    }
    for (int i = tid; i < kRows * P; i += nthreads) reinterpret_cast<float*>(smem + neg_base)[i] = -__builtin_inff();
    if (tid < 4) sh_misc[tid] = (tid == 3) ? U - 1 : 0;   // [3]: the next utterance to score (from the last one down)
    __threadfence_block();
    __syncthreads();
    if (NARROW && too_many) {   // (uniform) more than 31 labels beside the blank: the ring cannot hold this text
        fail(6);                // CTCFA_ST_TOO_MANY_LABELS
        return;
    }

    int t_end;
    {
        float bv = red_v[0];
        int bt = red_t[0];
        for (int q = 1; q < (nthreads >> 6); ++q) {   // (every wave of the workgroup took part, scorers included)
            const float ov = red_v[q];
            const int ot = red_t[q];
            if (ot != 0x7fffffff && (bt == 0x7fffffff || ov > bv || (ov == bv && ot < bt))) {
                bv = ov;
                bt = ot;
            }
        }
        t_end = __builtin_amdgcn_readfirstlane(bt);
        if (p.flags & 4u) t_end = T - 1;
    }
    SB_MARK(1);
    // frames the path never visits: t = 0 and everything after the end cell
    for (int t = (t_end >= 1 ? t_end + 1 : 0) + tid; t < T; t += nthreads) {
        cp[t] = 0.0f;
        cps[t] = 0.0f;
        if (st) st[t] = -2;
    }
    if (tid == 0 && t_end >= 1) {
        cp[0] = 0.0f;
        cps[0] = 0.0f;
        if (st) st[0] = -2;
    }
    if (want_seg) lds_barrier();   // (the scoring may start on these frames while the walk is still on its way)

    const int top0 = C - 1 + shift;   // the end cell's padded column
    if (t_end >= 1) {
        const int jstart = (t_end - 1) >> 5;
        const bool preamble = (p.flags & 2u) != 0;
        const bool gratis = (p.flags & 1u) != 0;
        const uint32_t my_slot = slot0 + (uint32_t)(wave * SLOT_BYTES);
        if (tid == 0) rec[2 * jstart] = top0;
        if (wave < NW) {   // (the scoring stays at priority 0)
            const int pr = (p.flags & kBtFlagLowPriority) ? 0 : p.prio;
            if (pr >= 3) __builtin_amdgcn_s_setprio(3);
            else if (pr == 2) __builtin_amdgcn_s_setprio(2);
            else if (pr == 1) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }

        // ---- staging: the 32 V floats of block jb are contiguous in lpz ----
        constexpr int NQ = kRows * P / 256;   // dwordx4 loads per lane that cover a block
        const int elems = kRows * V;
        const uint32_t magicV = ((1u << 20) + (uint32_t)V - 1u) / (uint32_t)V;   // n / V == (n * magicV) >> 20 for n < 4096
        float4 stg[NQ];
        int stg_n0 = 0;   // first element (of the segment's lpz) of the block in stg[]
        const int nmax = T * V - 4 > 0 ? T * V - 4 : 0;   // last place a dwordx4 load may start
        bool stg_blank_only = false;   // the block in stg[] lies in the start column: only its blank entries were asked for
        // GATHER: a staged row holds 32 entries picked from the vocabulary row -- a narrowed plan's ring, or (fewer than 32
        // entries, pitch 32) the row itself, its last entry repeated: the same four-by-four gather instead of a scatter of
        // 32 V contiguous floats (29 entries: 81 -> 76.5 us alone for config 3's shape, 76.5 with 32)
        const bool GATHER = NARROW || (P == 32 && V < 32);
        constexpr int LQ = P / 4;      // lanes that share a staged row (a quad of entries each)
        int gcol[4] = {0, 0, 0, 0};    // the vocabulary entries behind staged entries 4 (lane % LQ) + k
        if (GATHER) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = 4 * (lane % LQ) + k;
                gcol[k] = NARROW ? nrw[q] : (q < V ? q : V - 1);
            }
        }
        const bool gquad = GATHER && gcol[1] == gcol[0] + 1 && gcol[2] == gcol[0] + 2 && gcol[3] == gcol[0] + 3;
        auto issue = [&](int jb, bool blank_only = false) {
            const int n0 = (jb * kRows + 1) * V;
            stg_n0 = n0;
            stg_blank_only = blank_only;
            if (blank_only) {
                // the path sits in column 0 from here to the first frame: nothing to recompute, and the per-frame
                // outputs of such a block need the blank posterior of its 32 frames, not its 32 V emissions
                int t = jb * kRows + 1 + (lane & 31);
                t = t < T ? t : T - 1;
                stg[0].x = seg_lpz[t * V + p.blank];
                return;
            }
            if (GATHER) {   // quad q of lane l: staged entries 4 (l % LQ) .. + 3 of row l / LQ + (64 / LQ) q -- put()'s float4 layout at pitch P
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    int t = jb * kRows + 1 + lane / LQ + (64 / LQ) * q;
                    t = t < T ? t : T - 1;   // (rows past the end of the segment: its last row again, nobody reads them)
                    const float* __restrict__ rowp = seg_lpz + (size_t)t * (size_t)V;
                    if (gquad) {   // four neighbours of the vocabulary row: one load (a ring is sorted by vocabulary entry: the usual case)
                        const F4U v = *reinterpret_cast<const F4U*>(rowp + gcol[0]);
                        stg[q] = make_float4(v.x, v.y, v.z, v.w);
                    } else {
                        stg[q] = make_float4(rowp[gcol[0]], rowp[gcol[1]], rowp[gcol[2]], rowp[gcol[3]]);
                    }
                }
                return;
            }
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                int n = n0 + 4 * (lane + 64 * q);
                n = n < nmax ? n : nmax;   // past the end of the segment: its last four entries again (put() sorts them out)
                const F4U v = *reinterpret_cast<const F4U*>(seg_lpz + n);
                stg[q] = make_float4(v.x, v.y, v.z, v.w);
            }
        };
        auto put = [&]() {
            if (stg_blank_only) {
                if (lane < kRows) *reinterpret_cast<float*>(smem + my_slot + (uint32_t)((lane * P + rblank) * 4)) = stg[0].x;
                return;
            }
            if (GATHER || V == P) {
                // (rows past the end of the segment hold its last entries: nobody reads what becomes of them)
#pragma unroll
                for (int q = 0; q < NQ; ++q) *reinterpret_cast<float4*>(smem + my_slot + (uint32_t)((lane + 64 * q) * 16)) = stg[q];
            } else {
                int ln = lane;
                asm volatile("" : "+v"(ln));   // (keeps the 4 NQ element addresses out of registers between turns)
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    // a load that was moved back to nmax holds element n + k at component k + (n - nmax): matters when
                    // the vocabulary is not a multiple of four and a block's last valid row ends inside a quad
                    const int over = stg_n0 + 4 * (ln + 64 * q) - nmax;
                    const int sh = over > 0 ? over : 0;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int src = k + sh;
                        const float val = src == 0 ? stg[q].x : src == 1 ? stg[q].y : src == 2 ? stg[q].z : stg[q].w;
                        const uint32_t m = (uint32_t)(4 * (ln + 64 * q) + k);
                        const uint32_t row = (m * magicV) >> 20;
                        const uint32_t cv = m - row * (uint32_t)V;
                        // elements past the block (a pitch wider than the vocabulary): parked on the -inf column's last row, entry 1
                        const uint32_t ad = (int)m < elems ? my_slot + (row * P + cv) * 4u : neg_base + (uint32_t)(((kRows - 1) * P + 1) * 4);
                        *reinterpret_cast<float*>(smem + ad) = val;
                    }
                }
            }
        };

        // ---- one block: recurrence + residual test over the 64 columns below `top` ----
        // The walk is composed while the rows are computed, for EVERY lane the path might enter the block in:
        // after row k, Sv[l] holds the SWITCH bits (row 31 - b at bit b) of the path that is in lane l at row k,
        // down to the block's first row -- Sv_k[l] = switch(k, l) ? (Sv_{k-1} | bit k)[l + 1] : Sv_{k-1}[l], one v_or and
        // one v_cndmask with a wave_shl:1 DPP source per row.  When the block's true entry lane x arrives, the walk
        // of the block is one v_readlane (the path leaves in lane x + popcount): nothing of it is left on the chain
        // from block to block.
        auto recompute = [&](int j, int top, int ilast, uint32_t& Sout) {
            const int col = top - lane;
            const int c = col - shift;
            const bool pseudo = c <= 0;                    // start column and left of it: e = -inf
            const uint32_t widx = (j >= 1 && col >= 0) ? (uint32_t)((j - 1) * p.Cpad + col) : 0u;
            float prev = __uint_as_float(seg_bits[widx]);  // table row 32 j (issued first: the LDS reads below hide part of it)
            const int lab = pseudo ? 0 : (int)labs[c];     // c <= C-1: the path never sits right of the end cell's column
            const bool pp = pseudo && preamble;
            const uint32_t ea = pseudo ? neg_base : my_slot + (uint32_t)lab * 4u;
            const uint32_t la = pp ? neg_base : my_slot + (uint32_t)rblank * 4u;
            const float flo = pp ? 0.0f : kProbMax;        // the start column stays for free under preamble_transition_cost_zero
            // emission operands: a ring of RH rows in registers, row i + RH requested while row i is computed
            constexpr int RH = CTCFA_SB_RING;
            float e[RH], lb[RH];
#pragma unroll
            for (int i = 0; i < RH; ++i) {
                e[i] = *reinterpret_cast<const float*>(smem + ea + i * (P * 4));
                lb[i] = *reinterpret_cast<const float*>(smem + la + i * (P * 4));
            }
            if (j == 0) prev = pseudo ? 0.0f : kProbMax;   // table row 0
            else if (col < 0) prev = 0.0f;                 // left of the padded table (feeds nothing that is read)
            // m: the stay step the package's BACKTRACK assumes (max(blank, label)); mf: the one the FILL charged
            // -- 0 in a column labelled blank under blank_transition_cost_zero, else m
            const bool free_stay = gratis && !pseudo && lab == rblank;
            uint32_t Sv = 0u;
            auto rows = [&](auto gratis_tag, auto first_tag) {
                auto row = [&](auto row_tag) {
                    constexpr int i = decltype(row_tag)::value;   // (a constant the asm below can take as a literal)
                    if (decltype(first_tag)::value && i > ilast) return;   // the end cell's block: rows after it are not part of the path
                    const float ei = e[i % RH], li = lb[i % RH];
                    if (i % 2 == 1 && i + RH < kRows) {   // rows i - 1 and i are consumed: their ring entries take rows i + RH - 1, i + RH (one ds_read2_b32)
#pragma unroll
                        for (int r = i + RH - 1; r <= i + RH; ++r) {
                            e[r % RH] = *reinterpret_cast<const float*>(smem + ea + r * (P * 4));
                            lb[r % RH] = *reinterpret_cast<const float*>(smem + la + r * (P * 4));
                        }
                    }
                    const float m = max3f(li, ei, flo);
                    const float mf = (decltype(gratis_tag)::value && free_stay) ? 0.0f : m;
                    const float pl = dpp_wave_shl1(prev);
                    const float aa = pl + ei;
                    const float bb = prev + mf;
                    const float nw = max3f(aa, bb, kProbMax);
                    const float rsw = ei - (nw - pl);
                    const float rst = m - (nw - prev);
                    // SWITCH iff |rst| > |rsw| (ties and NaNs stay; a start-column lane has |rsw| = inf: it always stays).
                    // vcc = STAY; Sv = stay ? Sv : (Sv | bit)[l + 1] -- the lane shift rides on the v_cndmask as a DPP source
                    // modifier (the compiler emits a separate v_mov_dpp instead).  Wait states gfx950 wants and nobody
                    // inserts inside an asm statement: a VALU that reads VCC two after the VALU that wrote it; a DPP
                    // source two after its last VALU write (U) -- the v_or and the s_nop 1 stand in both gaps.
                    uint32_t U;
                    asm volatile(
                        "v_cmp_ngt_f32_e64 vcc, |%2|, |%3|\n\t"
                        "v_or_b32 %1, %4, %0\n\t"
                        "s_nop 1\n\t"
                        "v_cndmask_b32_dpp %0, %1, %0, vcc wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                        : "+v"(Sv), "=&v"(U)
                        : "v"(rst), "v"(rsw), "i"(1u << (31 - i))
                        : "vcc");
                    prev = nw;
                };
                for_each_row(row, std::make_integer_sequence<int, kRows>{});
            };
            if (ilast < kRows - 1) {
                if (gratis) rows(std::true_type{}, std::true_type{});
                else rows(std::false_type{}, std::true_type{});
            } else {
                if (gratis) rows(std::true_type{}, std::false_type{});
                else rows(std::false_type{}, std::false_type{});
            }
            Sout = Sv;
        };

        int j = wave < NW ? jstart - wave : -1;
        if (j >= 0) {
            issue(j);
            put();
        }
        int spin_fail = 0;
#ifdef CTCFA_BT_STAMP
        s_last = (uint32_t)__builtin_amdgcn_s_memtime();
        sk[5] = s_last - sk[1];   // pre-loop + first block's rows from HBM
#endif
        for (; j >= 0; j -= NW) {
            SB_COUNT(13);
            // ---- anchor: the nearest block above whose entry column is known ----
            int top, d, top_anchor = 0;
            {
                int spins = 0;
                unsigned long long known;
                int val;
                do {
                    const int jj = j + lane;
                    val = (lane < NW && jj <= jstart) ? (int)rec[2 * jj] : kSbSentinel;
                    known = __builtin_amdgcn_ballot_w64(val != kSbSentinel);
                    if (known == 0ull) {
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > kSpinCap) break;
                    }
                } while (known == 0ull);
                if (known == 0ull) {   // (cannot happen: my own last walk, or the kernel start, published one)
                    spin_fail = 1;
                    d = 0;
                    top = 0;
                } else {
                    d = __builtin_ctzll(known);
                    const int E = __builtin_amdgcn_readlane(val, d);
                    top = E;
                    top_anchor = E > top0 ? top0 : E;
                    if (d > 0) {
                        // columns the path is expected to drop over the d blocks in between: the mean of its slope over
                        // the last blocks walked (up to 8) and the slope that takes it from here to the first frame
                        const int jj = j + d;
                        int hist = jstart - jj;
                        hist = hist > 8 ? 8 : hist;
                        const float togo = (float)(E - shift) * 256.0f * __builtin_amdgcn_rcpf((float)(jj + 1));
                        int drop_q8 = __builtin_amdgcn_readfirstlane((int)togo);
                        if (hist >= 2) {
                            const int Eh = (int)rec[2 * (jj + hist)];
                            const int local_q8 = (int)((float)(Eh - E) * 256.0f * __builtin_amdgcn_rcpf((float)hist));
                            drop_q8 = (drop_q8 + __builtin_amdgcn_readfirstlane(local_q8)) >> 1;
                        }
                        const int pd = (drop_q8 * d) >> 8;
                        const int lift = pd - CTCFA_SB_MARGIN;
                        if (lift > 0) top = E - lift;
                    }
                }
                // (whatever the words in LDS say, no window starts outside the padded table: every global address
                // below is derived from `top`)
                top = top < 0 ? 0 : (top > top0 ? top0 : top);
            }
            // ---- windows.  A wave that has recomputed its window and still has no entry column does not sit and
            // poll: it recomputes the window NEXT to it (32 columns further down, then 32 further up) -- every
            // window kept is one VGPR (Sv) and one SGPR (its top), and the path may then enter anywhere in 64 or
            // 96 columns around the prediction.  The time would have been spent waiting; what a miss costs is a
            // whole recompute on the chain from block to block.
            constexpr int kCand = 3;
            int ctop[kCand];
            uint32_t cS[kCand];
            int ncand = 0;
            uint32_t Svv = 0u;
            int x = 0;
            const int ilast = (j == jstart) ? ((t_end - 1) & 31) : kRows - 1;
            const int anchorE = top_anchor;
            const bool in_start_column = anchorE - shift <= 0;   // the path has reached column 0: it stays there, frame by frame
            if (j - NW >= 0) issue(j - NW, in_start_column);   // my next block's rows: a whole turn in flight
            SB_LAP(6);    // anchor
            if (in_start_column) {
                top = anchorE;   // == this block's entry column: no recurrence to run, no SWITCH to find
                d = 0;
            } else {
                int next_top = top;
                for (;;) {
                    uint32_t Snew;
                    recompute(j, next_top, ilast, Snew);
                    asm volatile("" :: "v"(Snew));
                    SB_LAP(7);    // recompute
                    if (d == 0) {   // exact window
                        top = next_top;
                        Svv = Snew;
                        x = 0;
                        break;
                    }
                    ctop[ncand < kCand ? ncand : kCand - 1] = next_top;
                    cS[ncand < kCand ? ncand : kCand - 1] = Snew;
                    if (ncand < kCand) ++ncand;
                    // the block's true entry column (published by the wave that walked block j + 1)
                    int ent = __builtin_amdgcn_readfirstlane((int)rec[2 * j]);
                    int spins = 0;
                    if (ent == kSbSentinel && ncand < kCand && ncand < p.windows) {
                        // not there yet: another window meanwhile (below the first one, then above it)
                        int cand = ncand == 1 ? ctop[0] - kRows : ctop[0] + kRows;
                        cand = cand < 0 ? 0 : (cand > anchorE ? anchorE : cand);
                        bool fresh = true;
#pragma unroll
                        for (int k = 0; k < kCand; ++k) fresh = fresh && !(k < ncand && ctop[k] == cand);
                        if (fresh) {
                            next_top = cand;
                            SB_COUNT(12);
                            continue;
                        }
                    }
                    while (ent == kSbSentinel) {
                        ent = __builtin_amdgcn_readfirstlane((int)rec[2 * j]);
                        if (++spins > kSpinCap) {
                            spin_fail = 1;
                            ent = ctop[0];
                            break;
                        }
                    }
                    ent = ent > top0 ? top0 : ent;
                    SB_LAP(8);    // waiting for the block's entry column
                    bool hit = false;
#pragma unroll
                    for (int k = 0; k < kCand; ++k) {
                        const int xk = ctop[k < ncand ? k : 0] - ent;
                        if (!hit && k < ncand && xk >= 0 && xk < kRows) {
                            hit = true;
                            top = ctop[k];
                            Svv = cS[k];
                            x = xk;
                        }
                    }
                    if (hit) break;
                    SB_COUNT(14);
                    next_top = ent < 0 ? 0 : ent;   // the path entered outside every window: once more, from the exact entry column
                    d = 0;
                }
            }
            // ---- the walk of this block: look up the entry lane ----
            const uint32_t S = in_start_column ? 0u : (uint32_t)__builtin_amdgcn_readlane((int)Svv, x);
            const int pos = x + __builtin_popcount(S);   // one lane down per SWITCH
            SB_LAP(9);    // walk
            const int entry = top - x;          // padded column in which the path enters block j (at its last row)
            const int leave = top - pos;        // ... and the one it is in when it reaches row 32 j
#ifdef CTCFA_BT_STAMP
            if (lane == 0 && j >= 1) ((volatile uint32_t*)sh_pubtime)[(j - 1) & 255] = (uint32_t)__builtin_amdgcn_s_memtime();
#endif
            if (lane == 0) {
                if (j >= 1) rec[2 * (j - 1)] = leave;   // first: the next walk waits for this word
                else {
                    sh_misc[0] = t_end;
                    sh_misc[1] = (leave - shift > 0);   // reached t == 0 in a label column: the package's IndexError
                }
            }
            // ---- per-frame outputs of this block, lanes = rows, from the emission rows still in my slot ----
            if (lane < kRows) {
                const int i = lane;
                const int t = j * kRows + 1 + i;
                if (t <= t_end) {
                    const int b = 31 - i;
                    const int pct = entry - __builtin_popcount(S & ((1u << b) - 1u));
                    const int sw = (S >> b) & 1u;
                    const int c = pct - shift;
                    const float lbv = *reinterpret_cast<const float*>(smem + my_slot + (uint32_t)((i * P + rblank) * 4));
                    float prob;
                    int s_lab = -1;
                    if (c <= 0) {
                        prob = __builtin_fmaxf(lbv, kMaxProb);
                    } else {
                        const int g = (int)labs[c];
                        const float ev = *reinterpret_cast<const float*>(smem + my_slot + (uint32_t)((i * P + g) * 4));
                        const float mx = __builtin_fmaxf(ev, kMaxProb);
                        if (sw) {
                            prob = mx;
                            s_lab = NARROW ? nrw[g] : g;   // (the state list names vocabulary entries)
                            fol[c] = t;
                            fol_lds[c] = t;
                        } else {
                            prob = (mx > lbv) ? mx : lbv;
                        }
                    }
                    cp[t] = prob;
                    cps[t] = prob;
                    if (st) st[t] = s_lab;
                }
            }
            if (lane == 0) rec[2 * j + 1] = 1;   // this block's frames are out (LDS executes a wave's operations in order)
            SB_LAP(10);   // publish + per-frame outputs
            if (j - NW >= 0) put();   // (waits for the loads issued at the top of this turn)
            SB_LAP(11);   // slot refill
        }
        SB_MARK(2);
        __builtin_amdgcn_s_setprio(0);
        // ---- determine_utterance_segments, from the last utterance down, by whichever wave has nothing else to do
        // (the scorer waves from the start, a strider once its blocks are walked).  Utterance u is ready when every
        // block from the one in which the path entered the column before its first label upwards has its frames out.
        if (want_seg) {
            const int32_t* ub = a.utt_begin + sd.utt_off + sd.seg_index;
            int jdone = jstart + 1;   // blocks jdone .. jstart are out
            for (;;) {
                int u = 0;
                if (lane == 0) u = __hip_atomic_fetch_add(&sh_misc[3], -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                u = __builtin_amdgcn_readfirstlane(u);
                if (u < 0) break;
                const int cneed = ub[u] - 1;
                for (int spins = 0;; ++spins) {
                    while (jdone > 0 && __builtin_amdgcn_readfirstlane((int)rec[2 * (jdone - 1) + 1]) != 0) --jdone;
                    if (jdone == 0) break;
                    const int tf = cneed >= 1 ? __builtin_amdgcn_readfirstlane(((volatile int32_t*)fol_lds)[cneed]) : 0;
                    if (tf > 0 && ((tf - 1) >> 5) >= jdone) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (spins > kSpinCap) {
                        spin_fail = 1;
                        break;
                    }
                }
                score_one_utterance(sd, p.L, p.dur, ub, fol_lds, cps, T, C, u, a.seg_start, a.seg_end, a.seg_score, lane);
            }
        }
        if (spin_fail && lane == 0) sh_misc[2] = 1;
    } else {
        if (tid == 0) {
            sh_misc[0] = t_end;
            sh_misc[1] = 1;   // t_end == 0 with C >= 2 label columns: the package's IndexError
        }
        // (no path: the backtrack fails below, nothing to score)
    }
    __threadfence_block();
    __syncthreads();
    SB_MARK(3);
    if (sh_misc[2]) {
        fail(5);
        return;
    }
    if (sh_misc[1]) {
        fail(2);
        return;
    }
    if (tid == 0) {
        a.status_out[sd.seg_index] = 0;
        a.t_end_out[sd.seg_index] = t_end;
    }
#ifdef CTCFA_BT_STAMP
    SB_MARK(4);
    if (stamp_out && lane == 0 && T * 4 >= kSbMaxWaves * 16 * 8)
        for (int k = 0; k < 16; ++k) stamp_out[wave * 16 + k] = k >= 1 && k <= 4 ? (unsigned long long)(sk[k] - sk[0]) : (unsigned long long)sk[k];
#endif
}
#undef SB_LAP
#undef SB_MARK
#undef SB_COUNT

// ---------------------------------------------------------------------------------------
// score_min_mean_over_L above 128 frames (2.6 s; the reference passes 30): the backtrack kernels keep their fixed
// 128-element summation and are given L = 128; this kernel then scores the utterances of every aligned segment
// again with the caller's L, from frame_of_label and char_prob where the backtrack left them in HBM.  Rare option:
// written to be right (NumPy's summation order at any length), not fast.  grid = segments, block = 256.
// ---------------------------------------------------------------------------------------
constexpr int kRescoreThreads = 256;
__global__ void __launch_bounds__(kRescoreThreads)
rescore_kernel(const SegDesc* __restrict__ segs, const int32_t* __restrict__ utt_begin, const int32_t* __restrict__ frame_of_label,
               const float* __restrict__ char_prob, int L, double dur, double* __restrict__ seg_start, double* __restrict__ seg_end,
               double* __restrict__ seg_score, const int32_t* __restrict__ status_out) {
    const SegDesc sd = segs[blockIdx.x];
    if (sd.U <= 0 || status_out[sd.seg_index] != 0) return;   // (failed segments keep their zeros)
    score_utterances<kRescoreThreads, NoTick, true>(sd, L, dur, utt_begin + sd.utt_off + sd.seg_index, frame_of_label + sd.lab_off,
                                                    char_prob + sd.frm_off, sd.T, sd.C, sd.U, seg_start, seg_end, seg_score);
}

// ---------------------------------------------------------------------------------------
// Wide vocabularies (sub-word models: hundreds to thousands of entries) through the staged kernels: a segment
// only ever looks at the emissions of its OWN labels and of the blank.  This pre-pass gathers those columns --
// at most 128 per emission block -- into a compact matrix [T, Vc] (column 0 = blank, then the block's distinct
// labels in ascending order, the rest copies of the blank column); labels are renumbered on the host.
// grid = (emission blocks, row chunks), block = 256 threads.
// ---------------------------------------------------------------------------------------
struct CompactBlock {
    int64_t src_off;   // elements into the wide emissions
    int64_t dst_off;   // elements into the compact matrix
    int32_t T;
    int32_t reserved;
};

__global__ void __launch_bounds__(256)
compact_kernel(const CompactBlock* __restrict__ blocks, const float* __restrict__ lpz, const int32_t* __restrict__ orig,
               int V, int Vc, float* __restrict__ out) {
    const CompactBlock b = blocks[blockIdx.x];
    const int32_t* __restrict__ cols = orig + (int64_t)blockIdx.x * Vc;
    const int64_t n = (int64_t)b.T * Vc;
    for (int64_t idx = (int64_t)blockIdx.y * blockDim.x + threadIdx.x; idx < n; idx += (int64_t)gridDim.y * blockDim.x) {
        const int64_t t = idx / Vc;
        const int k = (int)(idx - t * Vc);
        out[b.dst_off + idx] = lpz[b.src_off + t * V + cols[k]];
    }
}

// ---------------------------------------------------------------------------------------
// Windowed regime (T > min_window_size): ctc-segmentation keeps only W = min(window, T) rows
// per column; column c's window starts offsets[c] frames into the audio and the step from
// offsets[c-1] depends on where column c-1 had its (first) maximum.  That makes the fill
// sequential column after column, and inside a column the fp32 chain
//     x_t = max( b_t , x_{t-1} + m_t )
// is sequential by rounding.  One workgroup per segment; wave 0 fills: lanes = 64 consecutive
// rows, b_t / m_t in parallel, the chain as 64 dependent (v_add_dpp wave_shr:1, v_max) pairs --
// after k iterations lanes <= k are final.  The full fp32 table goes to HBM (column-major,
// coalesced), exactly what the package allocates, because its backtrack reads table rows
// through NumPy index wrap-around when a window was too small -- the walk (lane 0) restates
// that loop literally, including the IndexError that doubles the window.  Rare path (windows
// longer than 160 s): written for exactness, not speed.
// dynamic LDS: T floats (previous/current column in place during the fill, char_probs after).
// ---------------------------------------------------------------------------------------
constexpr int kWinThreads = 256;

struct WinParams {
    int V, blank;
    uint32_t flags;
    int L;
    int min_window, max_window;
    int lds_bytes;  // dynamic LDS given to the kernel
    int fast_walk;  // single labels: 64 steps of the walk at a time where the cells are plain (0: the literal step only)
    int prefill;    // band_fill_kernel ran before this kernel: a window whose control word says so has its table (by absolute frame) and offsets
    int S;          // label width: ground_truth_mat[:, 0..S-1] (tokens of 1..S characters ending in a column); 1 = single labels
    double dur;
};
constexpr int kMaxSpan = 16;   // widest label matrix the windowed kernel takes

__device__ __forceinline__ int64_t np_index(int64_t i, int64_t n, int& err) {
    if (i < 0) i += n;
    if (i < 0 || i >= n) {
        err = 1;
        return 0;
    }
    return i;
}

// ---------------------------------------------------------------------------------------
// Windowed regime, the fill done row by row (round 4).  In ABSOLUTE frames the package's windowed table is the plain
// recurrence restricted to a band: column c lives in frames [O_c, O_c + W), cells outside it count as -1e9 -- and O_c follows
// from the first maximum of column c - 1 over ITS whole band, which is what makes cython_fill_table sequential column after
// column (11.8 ns a cell on one lane's dependent chain: 139 ms for a 190 s window).  But the offsets are nearly predictable
// (0 until the maximum passes the window's middle, then int((T - W) / C) + 1 until T - W is spent), and a table computed
// under GUESSED offsets proves them: if the offsets derived from its columns' maxima are the guessed ones, then by induction
// over the columns (column c only looks left) they are the package's.  So: every column at once, a row per step, a lane per
// K columns (the left neighbour's last value through DPP / an LDS word per wave and a barrier per row) --
//   pass 0: bands from a first guess (the higher offset from column 2 on until T - W is spent);
//   pass p: bands from the offsets derived from pass p - 1's maxima; derive again; equal -> done, else once more (each pass
//           gets at least one more column right; after kBandMaxPasses the literal kernel takes over).
// The table goes to HBM by absolute frame, row-major [T][C] (coalesced; the walk of windowed_kernel adds the offsets), every
// pass anew.  Per cell the expressions of windowed_kernel's literal fill, so the bits are the same.
// Label matrices (S > 1) and windows that double (an IndexError in the walk) stay with the literal fill.
// workspace control words, behind the window's offsets: [C] 1 = table filled this way for W = min(min_window, T), [C + 1] the
// last column's first maximum (window row).
// dynamic LDS: O [C] | first maxima by absolute frame [C] | 2 x 16 exchange words | kBandSlots emission rows
// ---------------------------------------------------------------------------------------
constexpr int kBandThreads = 1024;
constexpr int kBandMaxPasses = 8;
constexpr int kBandPF = 8;      // emission rows in flight (registers of the lanes that stage them)
constexpr int kBandSlots = 4;   // staged emission rows in LDS: row tau + 2 is written while row tau is computed
__host__ __device__ constexpr int band_lds_bytes(int C, int V) { return C * 8 + 2 * 16 * 4 + kBandSlots * V * 4; }

struct __attribute__((packed, aligned(4))) F2U { float x, y; };   // two consecutive floats, dword aligned

template <int K>
__global__ void __launch_bounds__(kBandThreads)
band_fill_kernel(const SegDesc* __restrict__ segs, const int32_t* __restrict__ win_list, const float* __restrict__ lpz,
                 const int32_t* __restrict__ labels, float* __restrict__ table_ws, int32_t* __restrict__ offs_ws, WinParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int sh_same;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;   // the widest window's columns / K, in whole waves (and at least V lanes: they stage the rows)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SegDesc sd = segs[win_list[blockIdx.x]];
    const int T = sd.T, C = sd.C, V = p.V;
    const float* __restrict__ seg_lpz = lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off;
    float* __restrict__ table = table_ws + sd.win_off;   // [T][C] by absolute frame
    int32_t* __restrict__ offsets = offs_ws + sd.wcol_off;
    lds_vint* Ob = (lds_vint*)smem;                       // [C] band starts of the pass
    lds_vint* amax = Ob + C;                              // [C] first maximum of each column, absolute frame
    float* xch = reinterpret_cast<float*>(smem + (size_t)C * 8);   // [2][16] last column of every wave, row before
    float* stage = xch + 32;                              // [kBandSlots][V] emission rows
    const bool preamble = (p.flags & 2u) != 0u;
    const bool gratis = (p.flags & 1u) != 0u;
    const float pm = kProbMax;
    const float ninf = -__builtin_inff();
    const int Wwin = p.min_window < T ? p.min_window : T;
    const float mean_offset = (float)((double)(T - Wwin) / (double)C);
    const int higher_offset = (int)mean_offset + 1;
    const int c0 = tid * K;
    const bool wave_live = (wave * 64 + 63) * K + K <= C;   // every column of this wave exists: stores need no masks
    const bool stager = wave * 64 < V;                      // this wave holds entries of the emission rows on their way to LDS
    uint32_t laddr[K];   // byte offset of the column's label in a staged row
    bool live[K], free_stay[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = c0 + k;
        live[k] = c < C;
        const int g = (live[k] && c > 0) ? seg_lab[c] : -1;
        laddr[k] = (uint32_t)(g >= 0 ? g : p.blank) * 4u;   // column 0 (and dead lanes) read the blank entry, unused
        free_stay[k] = gratis && g == p.blank;
    }
    if (tid == 0) {
        offsets[C] = 0;   // not (yet) filled this way
        offsets[C + 1] = -1;
        // the first guess: column 0's first maximum is its row 1 under preamble_transition_cost_zero (a column of zeros), so
        // column 1 starts with it; from there on every column's maximum taken to lie past the window's middle -- the higher
        // offset until T - W is spent.  (Right or wrong, a pass derives the next guess from what it computed.)
        int sum = 0;
        Ob[0] = 0;
        for (int c = 1; c < C; ++c) {
            int b = (T - Wwin) - sum;
            if (higher_offset < b) b = higher_offset;
            if (c >= 2) sum += b;
            Ob[c] = sum;
        }
    }
    __syncthreads();
    // Emission rows reach the lanes through LDS: lane v < V of the workgroup loads entry v of row tau + 2 + kBandPF while row
    // tau is computed and writes the one that has arrived (row tau + 2) to the ring -- ONE load instruction a row and staging
    // wave instead of K + 1 gathers per lane.
    auto row_entry = [&](int tau) -> float {
        const int f = tau < T ? tau : T - 1;
        return seg_lpz[(int64_t)f * V + (tid < V ? tid : 0)];
    };
    bool done = false;
    for (int pass = 0; pass < kBandMaxPasses && !done; ++pass) {
        const int W = Wwin;
        // A cell outside its column's band is held as -inf: whatever is added to it loses against the -1e9 floor of the
        // switch candidate and against the switch candidate itself, which is exactly what the package's range checks make of
        // it (the row before a band's first one: no stay candidate; a source cell outside the left column's band: p = -1e9)
        // -- the row loop carries ONE test per cell, "inside the band".
        int O[K];
        float prev[K], best_v[K];
        int best_t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            O[k] = live[k] ? Ob[c0 + k] : 0;
            prev[k] = ninf;
            best_v[k] = ninf;   // (a cell inside a band is >= -1e9 and never NaN: the first one always counts)
            best_t[k] = -1;
        }
        if (lane == 63) xch[wave] = ninf, xch[16 + wave] = ninf;
        if (tid < V) {
            stage[0 * V + tid] = row_entry(0);
            stage[1 * V + tid] = row_entry(1);
        }
        float ering[kBandPF];   // rows tau + 2 .. tau + 1 + kBandPF of this lane's entry, on their way
#pragma unroll
        for (int q = 0; q < kBandPF; ++q) ering[q] = stager ? row_entry(2 + q) : 0.0f;
        __syncthreads();
        const unsigned char* stage_b = reinterpret_cast<const unsigned char*>(stage);
        float ev[K], lbv;   // row tau's operands, read from the ring a row ahead
        lbv = stage[p.blank];
#pragma unroll
        for (int k = 0; k < K; ++k) ev[k] = *reinterpret_cast<const float*>(stage_b + laddr[k]);
        for (int tau0 = 0; tau0 < T; tau0 += kBandPF) {
#pragma unroll
            for (int q = 0; q < kBandPF; ++q) {
                const int tau = tau0 + q;
                if (tau >= T) break;   // uniform
                // the column left of this lane's first one, a row ago: the lane before (DPP), the wave before (LDS)
                float left0 = dpp_wave_shr1(ninf, prev[K - 1]);
                if (wave > 0 && lane == 0) left0 = xch[((tau + 1) & 1) * 16 + wave - 1];
                // next row's operands (staged a row ago; the barrier at the end of that row made them visible)
                const unsigned char* nrow = stage_b + (uint32_t)(((tau + 1) & (kBandSlots - 1)) * V * 4);
                const float lb_next = *reinterpret_cast<const float*>(nrow + p.blank * 4);
                float ev_next[K];
#pragma unroll
                for (int k = 0; k < K; ++k) ev_next[k] = *reinterpret_cast<const float*>(nrow + laddr[k]);
                float nx[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float pin = k == 0 ? left0 : prev[k > 0 ? k - 1 : 0];
                    const float b = __builtin_fmaxf(pin + ev[k], pm);   // switch_prob = max(prob_max, p)
                    const float mlpz = __builtin_fmaxf(ev[k], pm);      // max_lpz_prob
                    float m = mlpz > lbv ? mlpz : lbv;                  // max(lpz[blank], max_lpz_prob), the package's order
                    if (gratis) m = free_stay[k] ? 0.0f : m;            // blank_transition_cost_zero
                    nx[k] = __builtin_fmaxf(prev[k] + m, b);
                }
                if (wave == 0) {   // column 0: no switch into it (table[0, 0] = 0), its own stay step
                    const float m0 = preamble ? 0.0f : (pm > lbv ? pm : lbv);
                    const float x0 = __builtin_fmaxf(prev[0] + m0, tau == 0 ? 0.0f : pm);
                    nx[0] = tid == 0 ? x0 : nx[0];
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool inb = (unsigned)(tau - O[k]) < (unsigned)W;
                    const float x = inb ? nx[k] : ninf;
                    prev[k] = x;
                    const bool up = x > best_v[k];   // (first maximum: strict)
                    best_v[k] = up ? x : best_v[k];
                    best_t[k] = up ? tau : best_t[k];
                }
                if (tau == 0 && tid == 0) {   // the package's loop over column 0 starts at row 1
                    best_v[0] = ninf;
                    best_t[0] = -1;
                }
                if (lane == 63) xch[(tau & 1) * 16 + wave] = prev[K - 1];
                {   // (cells outside a band go out as -inf: nobody reads them)
                    float* out = table + (int64_t)tau * C + c0;
                    if (wave_live) {
                        if constexpr (K % 4 == 0) {
#pragma unroll
                            for (int k = 0; k < K; k += 4) *reinterpret_cast<F4U*>(out + k) = F4U{prev[k], prev[k + 1], prev[k + 2], prev[k + 3]};
                        } else if constexpr (K % 2 == 0) {
#pragma unroll
                            for (int k = 0; k < K; k += 2) *reinterpret_cast<F2U*>(out + k) = F2U{prev[k], prev[k + 1]};
                        } else {
#pragma unroll
                            for (int k = 0; k < K; ++k) out[k] = prev[k];
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            if (live[k]) out[k] = prev[k];
                    }
                }
                if (stager) {
                    if (tid < V) stage[((tau + 2) & (kBandSlots - 1)) * V + tid] = ering[q];
                    ering[q] = row_entry(tau + 2 + kBandPF);
                }
                lbv = lb_next;
#pragma unroll
                for (int k = 0; k < K; ++k) ev[k] = ev_next[k];
                lds_barrier();   // (orders LDS traffic only: __syncthreads would wait for the loads just issued)
            }
        }
        // the offsets these maxima lead to
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (live[k]) amax[c0 + k] = best_t[k];
        __syncthreads();
        if (tid == 0) {
            int same = 1;
            int sum = 0;
            for (int c = 1; c < C; ++c) {
                const int am = amax[c - 1];
                int a = (am < 0 ? -1 : am - sum) - Wwin / 2;        // lastArgMax of column c - 1 is a row of ITS window
                if (a < 0) a = 0;
                int b = (T - Wwin) - sum;
                if (higher_offset < b) b = higher_offset;
                sum += a < b ? a : b;
                same &= Ob[c] == sum;
                Ob[c] = sum;
            }
            sh_same = same;
        }
        __syncthreads();
        done = sh_same != 0;
    }
    if (done) {
        for (int c = tid; c < C; c += nthreads) offsets[c] = Ob[c];
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            offsets[C + 1] = amax[C - 1] < 0 ? -1 : amax[C - 1] - Ob[C - 1];
            offsets[C] = 1;
        }
    }
}

// ---------------------------------------------------------------------------------------
// band_fill_halo_kernel: the same fill with a workgroup barrier every kBandRows rows instead of every row (vocabularies of
// at most 256 entries).  As in fill_kernel, the first HL = kBandRows / K lanes of a wave are a HALO -- copies of the wave
// before's last kBandRows columns, wrong by one more column per row (their own left neighbour is missing), set right from
// an LDS exchange area every kBandRows rows; a wave's own columns never see the error.  Emission rows come in blocks of
// kBandRows: the first V lanes load a block ahead into registers and write it to the other half of a two-block LDS ring
// before the barrier.  56 of 64 lanes carry columns of their own (K = 2): 12 waves for a window of 1 242 columns, and the
// row loop runs at the pace of its instructions (~450 cycles a row) instead of the barrier's (~880).
// dynamic LDS: O [C] | first maxima [C] | 2 x 16 x kBandRows exchange floats | 2 x kBandRows emission rows
// ---------------------------------------------------------------------------------------
constexpr int kBandRows = 16;
__host__ __device__ constexpr int band_halo_lds_bytes(int C, int V) { return C * 8 + 2 * 16 * kBandRows * 4 + 2 * kBandRows * V * 4; }
__host__ __device__ constexpr int band_halo_own_cols(int K) { return (64 - kBandRows / K) * K; }   // columns of its own per wave

template <int K>
__global__ void __launch_bounds__(kBandThreads)
band_fill_halo_kernel(const SegDesc* __restrict__ segs, const int32_t* __restrict__ win_list, const float* __restrict__ lpz,
                      const int32_t* __restrict__ labels, float* __restrict__ table_ws, int32_t* __restrict__ offs_ws, WinParams p) {
    static_assert(kBandRows % K == 0, "a halo is a whole number of lanes");
    constexpr int R = kBandRows;
    constexpr int HL = R / K;             // halo lanes
    constexpr int U = (64 - HL) * K;      // columns of its own per wave
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int sh_same;
    const int tid = threadIdx.x;
    const int nthreads = blockDim.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SegDesc sd = segs[win_list[blockIdx.x]];
    const int T = sd.T, C = sd.C, V = p.V;
    const float* __restrict__ seg_lpz = lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off;
    float* __restrict__ table = table_ws + sd.win_off;   // [T][C] by absolute frame
    int32_t* __restrict__ offsets = offs_ws + sd.wcol_off;
    lds_vint* Ob = (lds_vint*)smem;
    lds_vint* amax = Ob + C;
    float* xch = reinterpret_cast<float*>(smem + (size_t)C * 8);   // [2][16][R]: the last R columns of every wave at a block's end
    float* stage = xch + 2 * 16 * R;                               // [2][R][V] emission rows, a block per half
    const bool preamble = (p.flags & 2u) != 0u;
    const bool gratis = (p.flags & 1u) != 0u;
    const float pm = kProbMax;
    const float ninf = -__builtin_inff();
    const int Wwin = p.min_window < T ? p.min_window : T;
    const float mean_offset = (float)((double)(T - Wwin) / (double)C);
    const int higher_offset = (int)mean_offset + 1;
    const int c0 = wave * U + (lane - HL) * K;   // (negative: wave 0's halo lanes, padding)
    const bool own = lane >= HL;
    const bool col0_lane = wave == 0 && lane == HL;
    const bool wave_live = wave * U + U <= C;    // every column of its own exists: stores need no masks
    const bool stager = wave * 64 < V;
    uint32_t laddr[K];
    bool live[K], free_stay[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = c0 + k;
        live[k] = c >= 0 && c < C;
        const int g = (live[k] && c > 0) ? seg_lab[c] : -1;
        laddr[k] = (uint32_t)(g >= 0 ? g : p.blank) * 4u;
        free_stay[k] = gratis && g == p.blank;
    }
    if (tid == 0) {
        offsets[C] = 0;
        offsets[C + 1] = -1;
        int sum = 0;   // the first guess (see band_fill_kernel)
        Ob[0] = 0;
        for (int c = 1; c < C; ++c) {
            int b = (T - Wwin) - sum;
            if (higher_offset < b) b = higher_offset;
            if (c >= 2) sum += b;
            Ob[c] = sum;
        }
    }
    __syncthreads();
    auto row_entry = [&](int tau) -> float {
        const int f = tau < T ? tau : T - 1;
        return seg_lpz[(int64_t)f * V + (tid < V ? tid : 0)];
    };
    const int nblocks = (T + R - 1) / R;
    bool done = false;
    for (int pass = 0; pass < kBandMaxPasses && !done; ++pass) {
        const int W = Wwin;
        int O[K];
        float prev[K], best_v[K];
        int best_t[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            O[k] = live[k] ? Ob[c0 + k] : 0;
            prev[k] = ninf;
            best_v[k] = ninf;
            best_t[k] = -1;
        }
        // emission block 0 into half 0, block 1 on its way
        float ering[R];
        if (stager) {
#pragma unroll
            for (int r = 0; r < R; ++r) ering[r] = row_entry(r);
            if (tid < V) {
#pragma unroll
                for (int r = 0; r < R; ++r) stage[r * V + tid] = ering[r];
            }
#pragma unroll
            for (int r = 0; r < R; ++r) ering[r] = row_entry(R + r);
        }
        __syncthreads();
        for (int blk = 0; blk < nblocks; ++blk) {
            const unsigned char* half = reinterpret_cast<const unsigned char*>(stage) + (uint32_t)((blk & 1) * R * V * 4);
            float ev[K], lbv;
            lbv = *reinterpret_cast<const float*>(half + p.blank * 4);
#pragma unroll
            for (int k = 0; k < K; ++k) ev[k] = *reinterpret_cast<const float*>(half + laddr[k]);
#pragma unroll 2   // (all 16 rows unrolled, the compiler hoists their LDS reads and spills)
            for (int r = 0; r < R; ++r) {
                const int tau = blk * R + r;
                // next row's operands (the block's last row reads its own again: unused)
                const unsigned char* nrow = half + (uint32_t)((r + 1 < R ? r + 1 : r) * V * 4);
                const float lb_next = *reinterpret_cast<const float*>(nrow + p.blank * 4);
                float ev_next[K];
#pragma unroll
                for (int k = 0; k < K; ++k) ev_next[k] = *reinterpret_cast<const float*>(nrow + laddr[k]);
                const float left0 = dpp_wave_shr1(ninf, prev[K - 1]);   // (lane 0: nothing to its left -- the halo's growing error)
                float nx[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const float pin = k == 0 ? left0 : prev[k > 0 ? k - 1 : 0];
                    const float b = __builtin_fmaxf(pin + ev[k], pm);
                    float mlpz;   // max(e, -1e9): the bare instruction (fmaxf quiets its operand first: one more per cell)
                    asm("v_max_f32 %0, %1, %2" : "=v"(mlpz) : "v"(ev[k]), "v"(pm));
                    float m = mlpz > lbv ? mlpz : lbv;
                    m = free_stay[k] ? 0.0f : m;   // blank_transition_cost_zero (free_stay is false without the flag)
                    nx[k] = __builtin_fmaxf(prev[k] + m, b);
                }
                if (wave == 0) {   // column 0: no switch into it (table[0, 0] = 0), its own stay step
                    const float m0 = preamble ? 0.0f : (pm > lbv ? pm : lbv);
                    const float x0 = __builtin_fmaxf(prev[0] + m0, tau == 0 ? 0.0f : pm);
                    nx[0] = col0_lane ? x0 : nx[0];
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const bool inb = live[k] && (unsigned)(tau - O[k]) < (unsigned)W;   // (padding left of column 0 stays -inf)
                    const float x = inb ? nx[k] : ninf;
                    prev[k] = x;
                    const bool up = x > best_v[k];
                    best_v[k] = up ? x : best_v[k];
                    best_t[k] = up ? tau : best_t[k];
                }
                if (tau == 0 && col0_lane) {   // the package's loop over column 0 starts at row 1
                    best_v[0] = ninf;
                    best_t[0] = -1;
                }
                if (tau < T) {   // (uniform; the last block may run past the window: nothing kept)
                    float* out = table + (int64_t)tau * C + c0;
                    if (wave_live) {
                        if (own) {
                            if constexpr (K % 4 == 0) {
#pragma unroll
                                for (int k = 0; k < K; k += 4) *reinterpret_cast<F4U*>(out + k) = F4U{prev[k], prev[k + 1], prev[k + 2], prev[k + 3]};
                            } else if constexpr (K % 2 == 0) {
#pragma unroll
                                for (int k = 0; k < K; k += 2) *reinterpret_cast<F2U*>(out + k) = F2U{prev[k], prev[k + 1]};
                            } else {
#pragma unroll
                                for (int k = 0; k < K; ++k) out[k] = prev[k];
                            }
                        }
                    } else {
#pragma unroll
                        for (int k = 0; k < K; ++k)
                            if (own && live[k]) out[k] = prev[k];
                    }
                }
                lbv = lb_next;
#pragma unroll
                for (int k = 0; k < K; ++k) ev[k] = ev_next[k];
            }
            // block end: the last R columns of every wave for the wave after it, the next emission block for everybody
            if (lane >= 64 - HL) {
#pragma unroll
                for (int k = 0; k < K; ++k) xch[((blk & 1) * 16 + wave) * R + (lane - (64 - HL)) * K + k] = prev[k];
            }
            if (stager) {
                if (tid < V) {
                    float* nh = stage + ((blk + 1) & 1) * R * V;
#pragma unroll
                    for (int r = 0; r < R; ++r) nh[r * V + tid] = ering[r];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) ering[r] = row_entry((blk + 2) * R + r);
            }
            lds_barrier();
            if (wave > 0 && lane < HL) {
#pragma unroll
                for (int k = 0; k < K; ++k) prev[k] = xch[((blk & 1) * 16 + wave - 1) * R + lane * K + k];
            }
        }
        // (a halo lane's maxima are its neighbour's business)
#pragma unroll
        for (int k = 0; k < K; ++k)
            if (own && live[k]) amax[c0 + k] = best_t[k];
        __syncthreads();
        if (tid == 0) {
            int same = 1;
            int sum = 0;
            for (int c = 1; c < C; ++c) {
                const int am = amax[c - 1];
                int a = (am < 0 ? -1 : am - sum) - Wwin / 2;
                if (a < 0) a = 0;
                int b = (T - Wwin) - sum;
                if (higher_offset < b) b = higher_offset;
                sum += a < b ? a : b;
                same &= Ob[c] == sum;
                Ob[c] = sum;
            }
            sh_same = same;
        }
        __syncthreads();
        done = sh_same != 0;
    }
    if (done) {
        for (int c = tid; c < C; c += nthreads) offsets[c] = Ob[c];
        __threadfence();
        __syncthreads();
        if (tid == 0) {
            offsets[C + 1] = amax[C - 1] < 0 ? -1 : amax[C - 1] - Ob[C - 1];
            offsets[C] = 1;
        }
    }
}

__global__ void __launch_bounds__(kWinThreads)
windowed_kernel(const SegDesc* __restrict__ segs, const int32_t* __restrict__ win_list,
                const float* __restrict__ lpz, const int32_t* __restrict__ labels,
                const int32_t* __restrict__ utt_begin, float* __restrict__ table_ws,
                int32_t* __restrict__ offs_ws, WinParams p, int32_t* __restrict__ frame_of_label,
                float* __restrict__ char_prob, int32_t* __restrict__ state,
                double* __restrict__ seg_start, double* __restrict__ seg_end,
                double* __restrict__ seg_score, int32_t* __restrict__ t_end_out,
                int32_t* __restrict__ status_out) {
    extern __shared__ __align__(16) unsigned char smem[];
    __shared__ int sh_res[2];  // [0] t_end, [1] error flag of the walk
    float* col = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const SegDesc sd = segs[win_list[blockIdx.x]];
    const int T = sd.T, C = sd.C, U = sd.U, V = p.V;
    const int S = p.S;   // label matrix [C, S] row-major, -1 padded (S == 1: the plain label sequence)
    const float* __restrict__ seg_lpz = lpz + sd.lpz_off;
    const int32_t* __restrict__ seg_lab = labels + sd.lab_off * S;
    int32_t* fol = frame_of_label + sd.lab_off;
    float* cp = char_prob + sd.frm_off;
    int32_t* st = state ? state + sd.frm_off : nullptr;
    float* table = table_ws + sd.win_off;      // column-major: table[c * W + t]
    int32_t* offsets = offs_ws + sd.wcol_off;  // [C]
    const bool want_seg = (utt_begin != nullptr) && (seg_score != nullptr) && U > 0;
    const bool preamble = (p.flags & 2u) != 0u;
    const float pm = kProbMax;

    long long window = p.min_window;
    int status = 0;
    int t_end = 0;
    for (;;) {
        const int W = (int)(window < (long long)T ? window : (long long)T);
        // outputs start from the package's initial state on every attempt
        for (int c = tid; c < C; c += kWinThreads) fol[c] = 0;
        for (int t = tid; t < T; t += kWinThreads) {
            cp[t] = 0.0f;
            if (st) st[t] = -2;
        }
        if (S > 1) {   // multi-character tokens read columns through Cython's index wrap-around: cells no
                       // column has reached yet must hold what `table.fill(config.max_prob)` left there
            for (int64_t i = tid; i < (int64_t)W * C; i += kWinThreads) table[i] = kMaxProb;
            __threadfence();
        }
        __syncthreads();
        // Two column buffers when they fit: wave 0 then never stores to HBM (vmcnt retires in
        // order on gfx9 -- its emission prefetch would wait for the previous chunk's table store in
        // every chunk) and waves 1..3 copy the finished column to the table while the next one is
        // being computed.  Otherwise one buffer, updated in place, and wave 0 stores itself.
        const bool dbl = (int64_t)W * 8 <= (int64_t)p.lds_bytes;
        // the first attempt of a single-label window: band_fill_kernel has filled the table (row-major by ABSOLUTE frame) and
        // the offsets, if its control word says so
        const bool pre = p.prefill && S == 1 && window == (long long)p.min_window &&
                         __hip_atomic_load(offsets + C, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1;
        // ------------------------------- fill ------------------------------------------
        const float mean_offset = (float)((double)(T - W) / (double)C);
        const int higher_offset = (int)mean_offset + 1;
        int offset_sum = 0, last_arg = -1;
        int cur_off[kMaxSpan];   // cur_offset[s] of cython_fill_table (wave-uniform)
#pragma unroll
        for (int q = 0; q < kMaxSpan; ++q) cur_off[q] = -1;
        const int nchunk = (W + 63) / 64;
        if (pre) last_arg = __hip_atomic_load(offsets + C + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int c = 0; c < (pre ? 0 : C); ++c) {
            float* colw = col + ((dbl && (c & 1)) ? W : 0);         // this column
            const float* colr = col + ((dbl && !(c & 1)) ? W : 0);  // previous column
            if (wave == 0) {
                int off = 0;
                if (c > 0) {
                    int a = last_arg - W / 2;
                    if (a < 0) a = 0;
                    int b = (T - W) - offset_sum;
                    if (higher_offset < b) b = higher_offset;
                    off = a < b ? a : b;
                    offset_sum += off;
#pragma unroll
                    for (int q = 0; q + 1 < kMaxSpan; ++q)   // in place, ascending, as the package does it
                        if (q + 1 < S) cur_off[q + 1] = cur_off[q] + off;
                    cur_off[0] = off;
                }
                if (lane == 0) __hip_atomic_store(offsets + c, offset_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int g = seg_lab[(int64_t)c * S];  // -1 for c == 0 (and where no one-character token ends)
                // first maximum of the column: every lane keeps the best of its own rows (ascending t,
                // strict '>' keeps the first), the lanes are reduced once per column
                float best_v = -__builtin_inff();
                int best_t = 0x7fffffff;
                float carry = 0.0f;  // x of the row before this chunk
                // previous column's value for the switch candidate of row t: col[t - 1 + off]
                auto load_prev = [&](int t0) -> float {
                    const int t = t0 + lane;
                    const int r = t - 1 + off;
                    return (c > 0 && t < W && r >= 0 && r < W) ? colr[r] : pm;
                };
                float pin = load_prev(0);
                float* tcol = table + (int64_t)c * W;
                // emissions of the next chunk are requested before this chunk's chain (HBM/L2 latency
                // would otherwise sit in front of every 64-row chain)
                auto load_em = [&](int t0, float& e_out, float& lb_out) {
                    const int t = t0 + lane;
                    const int f = (t < W) ? t + offset_sum : T - 1;
                    // raw loads only (any arithmetic on the results here would make the wave wait
                    // for them a chunk early); column 0 has no label: read the blank entry, unused
                    lb_out = seg_lpz[(int64_t)f * V + p.blank];
                    e_out = seg_lpz[(int64_t)f * V + ((c > 0 && g >= 0) ? g : p.blank)];
                };
                // Chunks go in groups of G: the emissions of the NEXT group are requested at the top of
                // a group (the compiler drains vmcnt at the loop's back edge, so a group has to be
                // long enough -- G chains -- to cover the gather latency).
                constexpr int G = 4;
                float e_cur[G], lb_cur[G], e_nxt[G], lb_nxt[G];
#pragma unroll
                for (int q = 0; q < G; ++q) load_em(q * 64, e_cur[q], lb_cur[q]);
                for (int ch0 = 0; ch0 < nchunk; ch0 += G) {
#pragma unroll
                    for (int q = 0; q < G; ++q) load_em((ch0 + G + q) * 64, e_nxt[q], lb_nxt[q]);
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        const int ch = ch0 + q;
                        if (ch >= nchunk) break;  // uniform
                        const int t0 = ch * 64;
                        const int t = t0 + lane;
                        const bool valid = t < W;
                        const float pin_next = load_prev(t0 + 64);  // read before this chunk overwrites col[]
                        const float lb = lb_cur[q];
                        float b, m;
                        if (c > 0) {
                            b = pm;              // switch_prob
                            float mlpz = pm;     // max_lpz_prob
                            if (g >= 0) {
                                const float e = e_cur[q];
                                const int r = t - 1 + off;
                                const float pcand = (r >= W || r < 0) ? pm : pin + e;
                                b = pcand > pm ? pcand : pm;           // switch_prob = max(prob_max, p)
                                mlpz = e > pm ? e : pm;
                            }
                            // tokens of s + 1 characters ending in this column: they leave column c - (s + 1)
                            // (an index below 0 wraps, as in Cython) in the same frame step
                            for (int sp = 1; sp < S; ++sp) {
                                const int gs = seg_lab[(int64_t)c * S + sp];
                                if (gs < 0) continue;   // uniform
                                const int f = valid ? t + offset_sum : T - 1;
                                const float es = seg_lpz[(int64_t)f * V + gs];
                                int co = -1;
#pragma unroll
                                for (int z = 1; z < kMaxSpan; ++z) co = (z == sp) ? cur_off[z] : co;
                                float ps;
                                if (t >= W - (co - 1) || t - 1 + co < 0) {
                                    ps = pm;
                                } else {
                                    int pcw = c - (sp + 1);
                                    if (pcw < 0) pcw += C;
                                    ps = __hip_atomic_load(table + (int64_t)pcw * W + (t - 1 + co), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT) + es;
                                }
                                if (valid) {
                                    b = ps > b ? ps : b;
                                    mlpz = es > mlpz ? es : mlpz;
                                }
                            }
                            m = mlpz > lb ? mlpz : lb;             // max(lpz[blank], max_lpz_prob)
                            if ((p.flags & 1u) && g == p.blank) m = 0.0f;   // blank_transition_cost_zero
                        } else {
                            b = (t == 0) ? 0.0f : pm;              // table[0,0] = 0; no switch into column 0
                            m = preamble ? 0.0f : (pm > lb ? pm : lb);
                        }
                        // chain: lane i <- max(b_i, x_{i-1} + m_i).  Row 0 has no stay candidate: m = -inf
                        // there makes the sum lose against b >= -1e9.  Two VALU per step: the add reads
                        // lane i-1 through DPP (lane 0 has no source lane and keeps carry + m_0).
                        if (t == 0) m = -__builtin_inff();
                        float tmp = carry + m;
                        float x = __builtin_fmaxf(tmp, b);
#pragma unroll 9
                        for (int it = 0; it < 63; ++it) {
                            // s_nop 1: a DPP operand written by the previous VALU needs two wait states,
                            // and the hazard recogniser does not look inside inline asm
                            asm volatile("s_nop 1\n\t"
                                         "v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf\n\t"
                                         "v_max_f32 %1, %0, %3"
                                         : "+v"(tmp), "+v"(x) : "v"(m), "v"(b));
                        }
                        if (valid) {
                            colw[t] = x;
                            if (!dbl) tcol[t] = x;
                        }
                        {   // rows t >= 1 only for column 0 (the package's loop starts there)
                            const bool counted = valid && !(c == 0 && t == 0);
                            if (counted && (best_t == 0x7fffffff || x > best_v)) {
                                best_v = x;
                                best_t = t;
                            }
                        }
                        const int lastl = (W - 1 - t0) < 63 ? (W - 1 - t0) : 63;
                        carry = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lastl));
                        pin = pin_next;
                    }
#pragma unroll
                    for (int q = 0; q < G; ++q) {
                        e_cur[q] = e_nxt[q];
                        lb_cur[q] = lb_nxt[q];
                    }
                }
#pragma unroll
                for (int o = 32; o >= 1; o >>= 1) {
                    const float ov = __shfl_xor(best_v, o);
                    const int ot = __shfl_xor(best_t, o);
                    if (ot != 0x7fffffff && (best_t == 0x7fffffff || ov > best_v || (ov == best_v && ot < best_t))) {
                        best_v = ov;
                        best_t = ot;
                    }
                }
                last_arg = (best_t == 0x7fffffff) ? -1 : best_t;
            }  // wave 0
            __syncthreads();
            if (dbl && wave != 0) {  // column c -> table, while wave 0 goes on with column c+1
                float* tcol = table + (int64_t)c * W;
                for (int t = tid - 64; t < W; t += kWinThreads - 64) tcol[t] = colw[t];
            }
        }
        __threadfence();
        __syncthreads();
        if (wave == 0) {
            // ------------------------------- walk -----------------------------------------
            // The package's loop, a step at a time, on lane 0 (`literal_step`) -- and, for plain labels away from the table's
            // edges, 64 steps at once: lane l looks at the cell l frames further back in the SAME column (as if the path had
            // stayed l times), every lane works out the package's two residuals from the same table entries and emissions
            // with the same fp32 operations, and the first lane that says SWITCH ends the run: the lanes before it are the
            // path's stays, it is the switch, the walk goes on one column to the left.  ~C + T / 64 rounds of loads instead of
            // T + C.  A lane whose cell is not plain (an index that would wrap or raise, a
            // residual that is not a number, an emission at or below max_prob) stops the run in front of it; if that is lane
            // 0 the literal step takes it.
            int err = 0;
            int te = (p.flags & 4u) ? W - 1 : last_arg;
            auto offs = [&](int64_t cc) -> int64_t {
                return (int64_t)__hip_atomic_load(offsets + cc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            auto tab = [&](int64_t r, int64_t cc) -> float {   // table[r, cc], r a row of column cc's window
                const int64_t at = pre ? (r + offs(cc)) * C + cc : cc * W + r;
                return __hip_atomic_load(table + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
            const double max_prob = -10000000000.0;
            int64_t t = te, c = C - 1, offset = 0;   // (wave-uniform)
            auto literal_step = [&]() {
                int min_s = -1;
                double min_delta = __builtin_inf();
                double max_lpz = max_prob;
                bool max_lpz_f32 = false;
                const int64_t cw = np_index(c, C, err);
                if (err) return;
                int g = -1;   // ground_truth[c, min_s]
                for (int sp = 0; sp < S && !err; ++sp) {
                    const int gs = seg_lab[cw * S + sp];
                    if (gs == -1) continue;
                    const int64_t pc = np_index(c - 1 - sp, C, err);
                    if (err) break;
                    offset = offs(cw) - ((c - sp > 0) ? offs(pc) : 0);   // (the last s looked at leaves its value behind, as in the package)
                    double sp_prob;
                    bool sp_f32 = false;
                    if (c > 0) {
                        const int64_t r = np_index(t + offs(cw), T, err);
                        const int64_t gi = np_index(gs, V, err);
                        if (err) break;
                        sp_prob = (double)seg_lpz[r * V + gi];
                        sp_f32 = true;
                    } else {
                        sp_prob = max_prob;
                    }
                    const int64_t r0 = np_index(t, W, err);
                    const int64_t r1 = np_index(t - 1 + offset, W, err);
                    if (err) break;
                    const float est32 = tab(r0, cw) - tab(r1, pc);
                    double delta;
                    if (sp_f32) delta = (double)__builtin_fabsf((float)sp_prob - est32);
                    else delta = __builtin_fabs(sp_prob - (double)est32);
                    if (delta < min_delta) {
                        min_delta = delta;
                        min_s = sp;
                        g = gs;
                    }
                    if (sp_prob > max_lpz) {
                        max_lpz = sp_prob;
                        max_lpz_f32 = sp_f32;
                    }
                }
                if (err) return;
                double stay;
                bool stay_f32 = false;
                if (t > 0) {
                    const int64_t r = np_index(t + offs(cw), T, err);
                    if (err) return;
                    const double lb = (double)seg_lpz[r * V + p.blank];
                    if (max_lpz > lb) {
                        stay = max_lpz;
                        stay_f32 = max_lpz_f32;
                    } else {
                        stay = lb;
                        stay_f32 = true;
                    }
                } else {
                    stay = max_prob;
                }
                const int64_t r0 = np_index(t, W, err);
                const int64_t r1 = np_index(t - 1, W, err);
                if (err) return;
                const float est_stay32 = tab(r0, cw) - tab(r1, cw);
                double stay_delta;
                if (stay_f32) stay_delta = (double)__builtin_fabsf((float)stay - est_stay32);
                else stay_delta = __builtin_fabs(stay - (double)est_stay32);
                if (stay_delta > min_delta) {
                    if (c > 0) {
                        const int64_t fr = np_index(offs(cw) + t, T, err);
                        if (err) return;
                        for (int sp = 0; sp <= min_s && !err; ++sp) {   // every character of the token
                            const int64_t ci = np_index(c - sp, C, err);
                            if (!err) fol[ci] = (int32_t)(offs(cw) + t);
                        }
                        if (err) return;
                        cp[fr] = (float)max_lpz;
                        if (st) st[fr] = g;
                    }
                    c -= 1 + min_s;
                    t -= 1 - offset;
                } else {
                    const int64_t fr = np_index(offs(cw) + t, T, err);
                    if (err) return;
                    cp[fr] = (float)stay;
                    if (st) st[fr] = -1;
                    t -= 1;
                }
            };
            int have_c = -1, have_Oc = 0, have_Op = 0;   // offsets of columns have_c and have_c - 1, loaded a round earlier
            // lanes that look ahead: 16 (a path stays T / C frames in a column, 7.6 in a 190 s window; every lane asks for two
            // table rows and an emission row of its own -- 64 lanes are 200 cache lines a round, ~10 us of one CU's misses),
            // all 64 after a round that met no switch (a long stay: silence)
            int ahead = 16;
            while ((t != 0 || c != 0) && !err) {
                bool taken = false;
                if (S == 1 && p.fast_walk && c >= 1 && t >= 1) {
                    const int Oc = have_c == (int)c ? have_Oc : (int)offs(c);
                    const int Op = have_c == (int)c ? have_Op : (int)offs(c - 1);
                    const int Onext = c >= 2 ? (int)offs(c - 2) : 0;   // (on its way while this round's cells are)
                    const int off = Oc - Op;
                    const int tl = (int)t - lane;
                    const int g = seg_lab[c];
                    const int64_t fr = (int64_t)tl + Oc;
                    bool reg = lane < ahead && tl >= 1 && tl < W && tl - 1 + off < W && fr < T && g >= 0 && g < V;
                    float spv = 0.0f, stay = 0.0f;
                    bool sw = false;
                    if (reg) {
                        // table[tl, c], table[tl - 1 + off, c - 1], table[tl - 1, c]: by absolute frame the two rows fr, fr - 1
                        auto cell = [&](int64_t r, int64_t cc, int64_t Occ) -> float {
                            const int64_t at = pre ? (r + Occ) * C + cc : cc * W + r;
                            return __hip_atomic_load(table + at, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        };
                        const float a = cell(tl, c, Oc), bl = cell(tl - 1 + off, c - 1, Op), d = cell(tl - 1, c, Oc);
                        spv = seg_lpz[fr * V + g];
                        const float lbv = seg_lpz[fr * V + p.blank];
                        const float delta = __builtin_fabsf(spv - (a - bl));          // fp32 - fp32, fp32 residual
                        stay = spv > lbv ? spv : lbv;                                  // max(lpz[blank], max_lpz_prob)
                        const float stay_delta = __builtin_fabsf(stay - (a - d));
                        reg = spv > -10000000000.0f && delta < __builtin_inff() && stay_delta == stay_delta;
                        sw = stay_delta > delta;
                    }
                    const unsigned long long bad = __builtin_amdgcn_ballot_w64(!reg);
                    const unsigned long long swm = __builtin_amdgcn_ballot_w64(reg && sw);
                    const int nreg = bad ? __builtin_ctzll(bad) : 64;
                    const int lsw = swm ? __builtin_ctzll(swm) : 64;
                    if (nreg > 0) {
                        taken = true;
                        const bool hit = lsw < nreg;
                        const int nst = hit ? lsw : nreg;   // lanes 0 .. nst - 1 are stays
                        if (lane < nst) {
                            cp[fr] = stay;
                            if (st) st[fr] = -1;
                        } else if (hit && lane == lsw) {
                            fol[c] = (int32_t)fr;
                            cp[fr] = spv;
                            if (st) st[fr] = g;
                        }
                        if (hit) {
                            t = t - lsw - 1 + off;
                            c -= 1;
                            offset = off;
                            have_c = (int)c;
                            have_Oc = Op;
                            have_Op = Onext;
                            ahead = 16;
                        } else {
                            ahead = 64;
                            t -= nreg;
                            have_c = (int)c;
                            have_Oc = Oc;
                            have_Op = Op;
                        }
                    }
                }
                if (S == 1 && p.fast_walk && !taken && c == 0 && t >= 1 && t < W && t < T && seg_lab[0] == -1) {
                    // The start column (no label: the package's loop over s finds nothing, min_delta stays inf and the step
                    // is a stay whatever the table holds) down to frame 0, every frame at once: char_probs = max(blank
                    // posterior, max_prob) in the package's order of comparison.  offsets[0] == 0.
                    for (int tt = (int)t - lane; tt >= 1; tt -= 64) {
                        const float lbv = seg_lpz[(int64_t)tt * V + p.blank];
                        cp[tt] = (-10000000000.0 > (double)lbv) ? -10000000000.0f : lbv;
                        if (st) st[tt] = -1;
                    }
                    t = 0;
                    taken = true;
                }
                if (!taken) {
                    if (lane == 0) literal_step();
                    t = __builtin_amdgcn_readfirstlane((int)t);
                    c = __builtin_amdgcn_readfirstlane((int)c);
                    offset = __builtin_amdgcn_readfirstlane((int)offset);
                    err = __builtin_amdgcn_readfirstlane(err);
                }
            }
            if (lane == 0) {
                sh_res[0] = te;
                sh_res[1] = err;
            }
        }
        __threadfence();
        __syncthreads();
        t_end = sh_res[0];
        const int err = sh_res[1];
        __syncthreads();
        if (!err) break;
        window *= 2;
        if (window < (long long)p.max_window) continue;
        status = 2;  // the package re-raises the IndexError
        break;
    }

    if (status != 0) {
        for (int c = tid; c < C; c += kWinThreads) fol[c] = 0;
        for (int t = tid; t < T; t += kWinThreads) {
            cp[t] = 0.0f;
            if (st) st[t] = -2;
        }
        if (want_seg)
            for (int u = tid; u < U; u += kWinThreads) {
                seg_start[sd.utt_off + u] = 0.0;
                seg_end[sd.utt_off + u] = 0.0;
                seg_score[sd.utt_off + u] = 0.0;
            }
        if (tid == 0) {
            status_out[sd.seg_index] = status;
            t_end_out[sd.seg_index] = -1;
        }
        return;
    }
    if (tid == 0) {
        status_out[sd.seg_index] = 0;
        t_end_out[sd.seg_index] = t_end;
    }
    if (!want_seg) return;
    // char_probs of this segment into LDS (the column buffer is free now)
    for (int t = tid; t < T; t += kWinThreads)
        col[t] = __hip_atomic_load(cp + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    score_utterances<kWinThreads>(sd, p.L, p.dur, utt_begin + sd.utt_off + sd.seg_index, fol, col, T, C, U,
                                  seg_start, seg_end, seg_score);
}

#endif  // CTCFA_FILL_KERNEL_ONLY
}  // namespace ctcfa
