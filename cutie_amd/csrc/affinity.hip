// Fused pixel-memory affinity readout (reference: MemoryManager.read -> get_similarity -> do_softmax(top_k)
// -> _readout; memory_manager.py:112-208, memory_utils.py:7-77).
//
// The reference materialises S[N,HW] in fp32, runs torch.topk over N, scatters the 30 weights back into a
// dense [N,HW] matrix and multiplies it with V[K*256,N].  Here the N x HW matrix never exists:
//
//   KEY_PREP     memory side (once per memorised frame): A_i = [k_i^2 | k_i] split into bf16 hi/lo, scale_i
//                query side  (per frame):                B_j = [-e_j | 2 k_j e_j] hi/lo, c_j = sum e_j k_j^2
//                so that S_ij = scale_i * (A_i . B_j - c_j) is ONE K=128 contraction.
//   AFF_SCORE/0  S tiles on MFMA (v_mfma_f32_16x16x32_bf16, 3 split terms hi*hi + hi*lo + lo*hi ~ fp32
//                accuracy), reduced to per-(16-token tile, query) maxima.
//   AFF_SELECT   tau_j = top_k-th largest tile maximum: a lower bound of the top_k-th largest score with
//                only ~top_k..1.2*top_k scores above it (radix select, exact).
//   AFF_SCORE/1  S tiles again (cheap: 3*2*128 flop per score), append the few scores >= tau_j.
//   AFF_READOUT  exact top-k of the candidates (ties -> lower slot), softmax, usage, sparse V gather.
//
// Memory tokens are addressed by *physical bank slot*; the bank is contiguous per bucket (keys) and per
// object (values), so the valid tokens are at most 3 slot ranges [long-term | permanent | working ring].
#include "common.h"
#include <stdlib.h>
#include <math.h>

__global__ void key_prep_kernel(const float* __restrict__ key, const float* __restrict__ aux, bf16_t* __restrict__ hi,
                                bf16_t* __restrict__ lo, float* __restrict__ sc, int n, int query) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)n * 16) return;
    int row = idx >> 4, c8 = idx & 15;                  // 16 chunks of 8 -> 128 operand channels
    // the lane's eight channels as two 16-byte loads, BEFORE any select on c8: `(c8 < 8) ? -e[i] : 2 k[i] e[i]` on pointers compiled to a
    // divergent branch per channel with the loads inside its arms -- eight dependent round trips (tools/isa_waits.py)
    const float* kp = key + (long)row * 64 + (c8 & 7) * 8;
    const float4 ka = *reinterpret_cast<const float4*>(kp), kb = *reinterpret_cast<const float4*>(kp + 4);
    const float k[8] = {ka.x, ka.y, ka.z, ka.w, kb.x, kb.y, kb.z, kb.w};
    float v[8];
    if (!(query & 1)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (c8 < 8) ? k[i] * k[i] : k[i];
        if (c8 == 0) sc[row] = aux[row] * 0.125f;       // shrinkage / sqrt(64)
    } else {
        const float* ep = aux + (long)row * 64 + (c8 & 7) * 8;
        const float4 ea = *reinterpret_cast<const float4*>(ep), eb = *reinterpret_cast<const float4*>(ep + 4);
        const float e[8] = {ea.x, ea.y, ea.z, ea.w, eb.x, eb.y, eb.z, eb.w};
        float pe[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { v[i] = (c8 < 8) ? -e[i] : 2.f * k[i] * e[i]; pe[i] = e[i] * k[i] * k[i]; }
        // c_j = sum_i e_i k_i^2, summed in channel order i = 0..63 as before -- but from the values the row's lanes hold anyway: lane c8 (< 8)
        // adds its eight products to the running sum of lane c8 - 1.  (One lane per row walking the 64 channels was 128 loads in a loop
        // that waits for each of them: tools/isa_waits.py; the row's 16 lanes sit in one wave, n * 16 threads exit as whole rows.)
        float c = 0.f;
        if (query & 2) {                                 // (A/B switch: the one-lane-per-row loop of rounds 1-3)
            if (c8 == 0) {
                const float* kk = key + (long)row * 64; const float* ee = aux + (long)row * 64;
                for (int i = 0; i < 64; ++i) c += ee[i] * kk[i] * kk[i];
                sc[row] = c;
            }
        } else {
#pragma unroll
            for (int step = 0; step < 8; ++step) {
                const float cin = __shfl(c, (threadIdx.x & 48) | (step > 0 ? step - 1 : 0), 64);
                if (c8 == step) {
                    c = step > 0 ? cin : 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) c += pe[i];
                }
            }
            c = __shfl(c, (threadIdx.x & 48) | 7, 64);
            if (c8 == 0) sc[row] = c;
        }
    }
    uint32_t h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16_t h0 = f2bf(v[2 * i]), h1 = f2bf(v[2 * i + 1]);
        bf16_t l0 = f2bf(v[2 * i] - bf2f(h0)), l1 = f2bf(v[2 * i + 1] - bf2f(h1));
        h[i] = (uint32_t)h0 | ((uint32_t)h1 << 16);
        l[i] = (uint32_t)l0 | ((uint32_t)l1 << 16);
    }
    *reinterpret_cast<uint4*>(hi + (long)row * 128 + c8 * 8) = make_uint4(h[0], h[1], h[2], h[3]);
    *reinterpret_cast<uint4*>(lo + (long)row * 128 + c8 * 8) = make_uint4(l[0], l[1], l[2], l[3]);
}

struct ScoreParams {
    const bf16_t* Ahi; const bf16_t* Alo; const float* scale;
    const bf16_t* Bhi; const bf16_t* Blo; const float* c;
    float* gmax_or_tau; float* cand_val; int* cand_idx; int* count;
    int HW, HWp, nranges, rs[3], rn[3], G, cap, mode, tiles_per_block, Gld;
    int prio;                                            // 1: s_setprio 1 (a read-out the caller's stream waits for; the stacked look-ahead passes stay at 0)
    int gbase, grem;                                     // 4-tile groups per block row: row by owns gbase (+ 1 if by < grem) groups
    int HWpf;                                            // query rows per frame (HWp = frames x HWpf; row j is a real query iff j % HWpf < HW)
    // ABI 4, clips in lock step (flags&4): the stacked frames belong to nbanks different memory banks, frame e to bank e % nbanks -- (A_hi, A_lo,
    // scale) of bank b at tbl[3 b ..]; the token ranges are those of every bank (banks of clips in lock step have one schedule)
    const unsigned long long* tbl; int nbanks;
#ifdef AFF_TIMELINE
    unsigned long long* tl;                              // (diagnostic library only: where the cycle stamps go; p10 of the op)
#endif
};

typedef __attribute__((ext_vector_type(4))) unsigned int au32x4;
typedef const __attribute__((address_space(1))) au32x4* aff_gptr16;
typedef const __attribute__((address_space(4))) unsigned long long* aff_ctbl;
typedef const __attribute__((address_space(1))) float* aff_gf32;      // (a pointer loaded from memory has no address space: say global, or hipcc emits flat_load)

// grid (ceil(HWp/128), ceil(G/tiles_per_block)); wave w of the block owns queries j0 + 32w .. +31 (two MFMA column sets).
// The memory operands of 4 consecutive 16-token tiles (64 rows x [hi|lo] x 256 B = 32 KB) are staged ONCE per block in
// LDS (double-buffered, register prefetch of the next group) and shared by the 4 waves; rows are XOR-swizzled by
// (row & 15) so the ds_read_b128 fragment reads are conflict-free (each of the instruction's four 16-lane groups covers the 16 slots of
// the 256-B bank row once; the conflicts the PMC counted in rounds 3-4 were the staging STORES, see `sr8` below).  The B fragments of the wave's 32 queries stay in
// registers, so one set of 8 A-fragment reads feeds 24 MFMAs (with 16 queries per wave the loop was LDS-bound).
// mode 1 appends through wave-private LDS lists (ballot + mbcnt positions, no atomics in the loop; one dense burst of
// global atomics per wave at the end).
// candidate counters: one per query, each in its own 128-B line (the ~50 returning atomics per query of the 77k per
// frame otherwise all land on 51 cache lines and serialise in a couple of L2 channels: +23 us)
#define AFF_CSTRIDE 32
#define AFF_TG 4                                      // tiles per LDS group
#define AFF_LCAP 1024                                 // LDS candidate list entries per block ...
#define AFF_WCAP (AFF_LCAP / 4)                       // ... = 4 wave-private lists (a full list is flushed and starts again)
// The wave's list -> the global per-query lists.  Also the path of a FULL list (round 6): until then what did not fit went to the global lists entry
// by entry -- a returning atomic and a wait per hit inside the tile loop -- and on the bench clip a third of all candidates did (hits per wave and
// block: mean 42, 1 % over 976, maximum 1659; tools/top2_probe.py): those few waves were the launch (33 us against a floor of 9).
// ONE global atomic per query and flush: the entries are ranked inside their query by LDS atomics first (a query of a candidate-dense region
// collects ~370 candidates; one returning atomic each on its counter's cache line serialised in the L2: the pass ran at that line's pace).
// The hits of one 16 x 16 fragment (lane: 4 tokens of its query) -> the wave's list.  The wave owns its queries: positions come from a wave-private
// counter + lane prefixes, no atomics and no waits inside the MFMA loop.  A lane holds 0..4 hits: the prefix over the lanes is built from the THREE bit
// planes of that count (three ballots, mbcnt on each) instead of one ballot -> branch -> append round per token row (round 6: the four rounds per fragment
// were most of an executed tile's ~1700 cycles in candidate-dense regions, where every fragment of a wave has hits).
#define AFF_MBCNT(B) __builtin_amdgcn_mbcnt_hi((unsigned)((B) >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)(B), 0u))
#define AFF_APPEND_HITS(S, JQ, THR, VALID, TOK0, WQ0_, NWQ_)                                                           \
    {                                                                                                      \
        bool h_[4];                                                                                        \
        int nl_ = 0;                                                                                       \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) { h_[q] = (VALID) && (S)[q] >= (THR) && (S)[q] > -INFINITY; nl_ += h_[q] ? 1 : 0; } \
        const unsigned long long b0_ = __ballot((nl_ & 1) != 0), b1_ = __ballot((nl_ & 2) != 0), b2_ = __ballot((nl_ & 4) != 0); \
        const int nh_ = __popcll(b0_) + 2 * __popcll(b1_) + 4 * __popcll(b2_);                             \
        if (nh_) {                                                      /* wave-uniform */                  \
            if (wcount + nh_ > AFF_WCAP) { AFF_FLUSH_LIST(wcount, WQ0_, NWQ_); wcount = 0; }   /* a full list is flushed and starts again (nh_ <= AFF_WCAP) */ \
            int pos_ = wcount + AFF_MBCNT(b0_) + 2 * AFF_MBCNT(b1_) + 4 * AFF_MBCNT(b2_);                  \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                  \
                if (h_[q]) { wl_j[pos_] = (JQ); wl_idx[pos_] = (TOK0) + q; wl_val[pos_] = (S)[q]; ++pos_; } \
            wcount += nh_;                                                                                 \
        }                                                                                                  \
    }
#define AFF_FLUSH_LIST(N, WQ0, NWQ)                                                                        \
    {                                                                                                      \
        const int n_ = (N);                                                                                \
        int fj_[AFF_WCAP / 64], fr_[AFF_WCAP / 64];                                                        \
        _Pragma("unroll") for (int r_ = 0; r_ < AFF_WCAP / 64; ++r_) fj_[r_] = r_ * 64 + lane < n_ ? wl_j[r_ * 64 + lane] - (WQ0) : -1; \
        /* rank of every entry among the entries of its query: returning LDS atomics on the wave's own counters (zero between two flushes) */ \
        _Pragma("unroll") for (int r_ = 0; r_ < AFF_WCAP / 64; ++r_) fr_[r_] = fj_[r_] >= 0 ? atomicAdd(&wl_cnt[fj_[r_]], 1) : 0; \
        if (lane < (NWQ)) {                                             /* ONE returning global atomic per query with entries */ \
            const int c_ = wl_cnt[lane];                                                                   \
            wl_cnt[64 + lane] = c_ ? atomicAdd(&p.count[((WQ0) + lane) * AFF_CSTRIDE], c_) : 0;           \
            wl_cnt[lane] = 0;                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int r_ = 0; r_ < AFF_WCAP / 64; ++r_)                                       \
            if (fj_[r_] >= 0) {                                                                            \
                const int pos_ = wl_cnt[64 + fj_[r_]] + fr_[r_];                                           \
                const long j_ = (WQ0) + fj_[r_];                                                           \
                if (pos_ < p.cap) { p.cand_val[j_ * p.cap + pos_] = wl_val[r_ * 64 + lane]; p.cand_idx[j_ * p.cap + pos_] = wl_idx[r_ * 64 + lane]; } \
            }                                                                                              \
    }
#define AFF_LDS_BYTES (2 * 2 * 64 * 16 * 16 + AFF_LCAP * 12 + 16 + 2 * 64 * 4)
#undef AFF_LDS_BYTES
#define AFF_LDS_BYTES (2 * 2 * 64 * 16 * 16 + AFF_LCAP * 12 + 16 + 2 * 2 * 64 * 4 + 4 * 128 * 4)     // (... + per wave 64 entry counters and 64 list bases of a flush)
// AFF_MODE (compile-time pass): 0 = per-(tile, query) maxima, 1 = candidate lists.  The max-only pass carries neither the
// candidate-list code nor its registers (6 instead of 22 non-MFMA instructions per MFMA).
// Round-2 changes, all from the instruction mix (the loop was SALU / VALU-bound, not MFMA-bound):
//   * a wave stages ONE tile of the 4-tile group (16 rows x [hi|lo] x 256 B), so the token-range arithmetic (which of the <= 3
//     slot ranges a tile lies in) is wave-uniform: once per wave and group in SGPRs instead of 5 times per thread in VGPRs;
//   * -c_j is the accumulator's initial value and the padding rows carry an additive -inf next to a zero scale, so a score costs
//     one v_fma (was: compare, subtract, multiply, select);
//   * pass 1 skips every (16-token tile, 16-query set) whose pass-0 maximum is below the set's thresholds (flags & 1: the gmax
//     matrix of pass 0 precedes tau in memory): ~85 % of the MFMA work of the second pass.
// Optional timeline (diagnostic library only: -DAFF_TIMELINE, tools/aff_timeline.py): s_memtime stamps of every wave of the logical
// blocks 0 and nb / 2, parked in LDS (4 KB behind the maxima of mode 0: no VMEM traffic that would disturb the vmcnt waits) and copied
// to p.cand_val (unused in mode 0; the tool hands in a buffer) when the wave ends.  Record: [count, stamps (cycles << 8 | id) ...].
#ifdef AFF_TIMELINE
#define ATL_MAX 95                                        // (4 waves x 95 stamps x 8 B = 3040 B: two blocks with it still fit the CU's 160 KB of LDS in pass 1)
#define ATL_DECL(BASE) unsigned long long* const atl_ = reinterpret_cast<unsigned long long*>(BASE) + wave * ATL_MAX; int atn_ = 0;
#define ATL(ID) { if (atn_ < ATL_MAX) { const unsigned long long t_ = (__builtin_readcyclecounter() << 8) | (unsigned)(ID); if ((threadIdx.x & 63) == 0) atl_[atn_] = t_; ++atn_; } }
#define ATL_DUMP(LOGICAL, NB)                                                                              \
    {                                                                                                      \
        const int slot_ = (LOGICAL) == 0 ? 0 : (LOGICAL) == (NB) / 2 ? 1 : -1;                             \
        unsigned long long* const tl_ = p.tl ? p.tl : (mode == 0 ? reinterpret_cast<unsigned long long*>(p.cand_val) : nullptr); \
        if (slot_ >= 0 && tl_) {                                                                           \
            __builtin_amdgcn_s_waitcnt(0);                                                                 \
            unsigned long long* o_ = tl_ + ((long)slot_ * 4 + wave) * (ATL_MAX + 1);                       \
            if ((threadIdx.x & 63) == 0) o_[0] = 0x41540000ull | (unsigned)atn_;                           \
            for (int q_ = threadIdx.x & 63; q_ < atn_; q_ += 64) o_[1 + q_] = atl_[q_];                    \
        }                                                                                                  \
    }
#else
#define ATL_DECL(BASE)
#define ATL(ID) {}
#define ATL_DUMP(LOGICAL, NB) {}
#endif
// The hand-over of a staged group: LDS counter drained + raw barrier.  __syncthreads() also drains vmcnt, i.e. it waited every group for
// the write acknowledgement of the tile maxima stored just before it (pass 0) and for the NEXT group's operand loads issued at the top
// of the iteration (both passes: the prefetch never overlapped the barrier).  Registers fed by global loads are waited for by the
// compiler where AFF_STORE uses them.
#define AFF_SYNC() { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(15 | (3 << 14) | (7 << 4) | (0 << 8)); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
template <int AFF_NQ, int AFF_MODE, bool JT = false>  // 16-query column sets per wave (1 or 2); JT: one bank per stacked frame (see ScoreParams::tbl)
__global__ __launch_bounds__(256) void aff_score_kernel(ScoreParams p) {
    if (p.prio) __builtin_amdgcn_s_setprio(1);
    constexpr int mode = AFF_MODE;
    extern __shared__ __attribute__((aligned(16))) unsigned char aff_smem[];
    au32x4 (*lds)[2 * 64 * 16] = reinterpret_cast<au32x4 (*)[2 * 64 * 16]>(aff_smem);   // [buffer][hi/lo][row][16 chunks]
    int* l_j = reinterpret_cast<int*>(aff_smem + 2 * 2 * 64 * 16 * 16);
    int* l_idx = l_j + AFF_LCAP;
    float* l_val = reinterpret_cast<float*>(l_idx + AFF_LCAP);
    int* l_n = reinterpret_cast<int*>(l_val + AFF_LCAP);
    float (*lsc)[64] = reinterpret_cast<float (*)[64]>(l_n + 4);        // [buffer][row of the group]: scale_i (0 for padding rows)
    float (*lpad)[64] = lsc + 2;                                        // [buffer][row]: 0, or -inf for padding rows
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping (see conv_igemm.hip): consecutive logical blocks share the token chunk (query block fastest)
    int bx, by;
    const int nb = gridDim.x * gridDim.y;
    int logical;
    {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        by = logical / (int)gridDim.x;
        bx = logical - by * (int)gridDim.x;
    }
    // (timeline builds: mode 0 parks the stamps in the unused candidate-list area, mode 1 in 3040 B the launch adds behind the regular LDS)
    ATL_DECL(mode == 0 ? aff_smem + 2 * 2 * 64 * 16 * 16 + 4096 : aff_smem + AFF_LDS_BYTES)
    ATL(0)
    // JT: the bank of this block's queries (a block's 64 * AFF_NQ query rows lie inside ONE frame: the launcher insists on HWpf % (64 * AFF_NQ) == 0)
    const bf16_t* Ahi_ = p.Ahi; const bf16_t* Alo_ = p.Alo; const float* scale_ = p.scale;
    if constexpr (JT) {
        const int b_ = __builtin_amdgcn_readfirstlane(((bx * (64 * AFF_NQ)) / p.HWpf) % p.nbanks);      // (the division runs on the VALU: say that the result is uniform, or the table is read by vector loads)
        const aff_ctbl tb_ = (aff_ctbl)p.tbl + 3 * b_;        // (constant address space: scalar loads -- a plain global load may not be scalar in a kernel that stores)
        Ahi_ = reinterpret_cast<const bf16_t*>(tb_[0]); Alo_ = reinterpret_cast<const bf16_t*>(tb_[1]); scale_ = reinterpret_cast<const float*>(tb_[2]);
    }
    int* wl_j = l_j + wave * AFF_WCAP; int* wl_idx = l_idx + wave * AFF_WCAP; float* wl_val = l_val + wave * AFF_WCAP;
    int* const wl_cnt = reinterpret_cast<int*>(lpad + 2) + wave * 128;  // AFF_FLUSH_LIST: [64] entries per query, [64] list bases (wave-private)
    if (mode == 1) { wl_cnt[lane] = 0; }
    const int wq0 = bx * (64 * AFF_NQ) + wave * (16 * AFF_NQ);          // first query of this wave
    int wcount = 0;                                                     // wave-uniform fill of the wave's list
    int jq[AFF_NQ];                                                     // query column of this lane, per set
    bool jvalid[AFF_NQ];
    float ncj[AFF_NQ], thr[AFF_NQ];
    bf16x8 bh[AFF_NQ][4], bl[AFF_NQ][4];
    const bool skip = mode == 1 && (p.mode & 2);                        // pass-0 maxima available: skip tiles without candidates
    const float* tau_p = p.gmax_or_tau;
    const float* gmax_p = p.gmax_or_tau - (long)p.HWp * p.Gld;
    // c_j and tau_j: unconditional loads of a clamped row, looked at only behind the prologue's barriers.  (The conditional form --
    // `jvalid ? -p.c[jq] : 0` -- compiled to a divergent branch with its own s_waitcnt vmcnt(0) per query set: two (pass 1: four)
    // serialized global round trips in front of every block's first operand load.)
    float craw[AFF_NQ], traw[AFF_NQ];
#pragma unroll
    for (int u = 0; u < AFF_NQ; ++u) {
        jq[u] = bx * (64 * AFF_NQ) + wave * (16 * AFF_NQ) + u * 16 + l15;
        // several frames' queries in one launch (one read-out per BANK VERSION, memory_manager.py `prefetch_affinity_batch`): the operand
        // rows of frame f are [f * HWpf, f * HWpf + HW), the rest of each frame's HWpf rows is zero padding
        jvalid[u] = jq[u] < p.HWp && jq[u] % p.HWpf < p.HW;
        const int jc = jvalid[u] ? jq[u] : 0;
        craw[u] = p.c[jc];
        traw[u] = mode == 1 ? tau_p[jc] : 0.f;
    }
    const int g0 = AFF_TG * (by * p.gbase + min(by, p.grem));
    const int g1 = min(g0 + AFF_TG * (p.gbase + (by < p.grem ? 1 : 0)), p.G);
    const int T0 = (p.rn[0] + 15) >> 4, T1 = (p.nranges > 1) ? ((p.rn[1] + 15) >> 4) : 0;
    // staging: wave w stages tile w of the group.  Eight consecutive lanes take eight consecutive 16-B chunks of ONE row (a whole 128-B
    // line per lane octet in the global load, and -- ds_write_b128 is served in groups of 8 contiguous lanes on 32 banks -- eight
    // distinct 16-B slots of the swizzled LDS row per group: (c ^ row) & 7 runs through 0..7).  Rounds 1-4 gave a lane four
    // consecutive chunks of a row (lane = row * 4 + quarter): each store group hit every slot twice, the 2-way conflict the PMC showed
    // as SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.23 (profiles/r04_summary.json), and every load instruction touched 32 lines for
    // a quarter of their bytes.  Same LDS image, same reads, same bits.
    const int sr8 = lane >> 3, c8l = lane & 7;
    au32x4 st[8];
    float st_sc = 0.f;
    int st_nv = 0;
    f32x4 gq[AFF_NQ];                                                   // pass-0 maxima of the NEXT group's 4 tiles (mode 1 + skip)
#pragma unroll
    for (int u = 0; u < AFF_NQ; ++u) gq[u] = (f32x4){INFINITY, INFINITY, INFINITY, INFINITY};
    // token slot of the first row / number of valid rows of tile g (wave-uniform)
    auto tile_slot = [&](int g, int& slot0, int& nvalid) {
        int start, n, lt;
        if (g < T0) { lt = g; start = p.rs[0]; n = p.rn[0]; }
        else if (g < T0 + T1) { lt = g - T0; start = p.rs[1]; n = p.rn[1]; }
        else { lt = g - T0 - T1; start = p.rs[2]; n = p.rn[2]; }
        slot0 = start + lt * 16;
        nvalid = min(16, n - lt * 16);
    };
#define AFF_LOAD(GRP)                                                                                      \
    {                                                                                                      \
        int gt_ = (GRP) + wave;                                /* this wave's tile of the group */         \
        gt_ = gt_ < g1 ? gt_ : g1 - 1;                         /* clamp: rows of missing tiles are never used */ \
        int slot0_, nv_;                                                                                   \
        tile_slot(gt_, slot0_, nv_);                                                                       \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
            const int row_ = sr8 + 8 * (e & 1);                                                            \
            const long off_ = (long)(slot0_ + min(row_, nv_ - 1)) * 128 + (c8l + 8 * (e >> 1)) * 8;        \
            st[e] = JT ? *(aff_gptr16)(Ahi_ + off_) : *reinterpret_cast<const au32x4*>(p.Ahi + off_);       \
            st[4 + e] = JT ? *(aff_gptr16)(Alo_ + off_) : *reinterpret_cast<const au32x4*>(p.Alo + off_);   \
        }                                                                                                  \
        /* the per-token scale rides along: a global load inside the MFMA loop would make every tile wait for this whole */ \
        /* prefetch (vmcnt is in-order); unconditional load of a clamped row, selected afterwards */       \
        st_sc = JT ? ((aff_gf32)scale_)[slot0_ + min(l15, nv_ - 1)] : p.scale[slot0_ + min(l15, nv_ - 1)];   /* (looked at in AFF_STORE: a select here waits for it) */ \
        st_nv = nv_;                                                                                       \
        if (skip) {                                                                                        \
            _Pragma("unroll") for (int u = 0; u < AFF_NQ; ++u)                                             \
                gq[u] = *reinterpret_cast<const f32x4*>(gmax_p + (long)min(jq[u], p.HWp - 1) * p.Gld + (GRP)); \
        }                                                                                                  \
    }
#define AFF_STORE(BUF)                                                                                     \
    {                                                                                                      \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                    \
            const int r16_ = sr8 + 8 * (e & 1), row_ = wave * 16 + r16_;                                   \
            const int ch_ = (c8l + 8 * (e >> 1)) ^ r16_;                                                   \
            lds[BUF][row_ * 16 + ch_] = st[e];                                                             \
            lds[BUF][1024 + row_ * 16 + ch_] = st[4 + e];                                                  \
        }                                                                                                  \
        if (l4 == 0) { lsc[BUF][wave * 16 + l15] = l15 < st_nv ? st_sc : 0.f; lpad[BUF][wave * 16 + l15] = l15 < st_nv ? 0.f : -INFINITY; } \
    }
    f32x4 gcur[AFF_NQ];
    // Prologue order (round 4): the first memory group's loads are in flight BEFORE the query operand is pulled through LDS, so the
    // block pays one global round trip in front of its first MFMA instead of two (vmcnt is in-order: the LDS writes of the query
    // rows wait for both).  Same bits: only the order of independent loads changed.  -DAFF_OLD_PROLOGUE (diagnostic library) keeps the
    // old order for the in-box A/B.
#ifndef AFF_OLD_PROLOGUE
    AFF_LOAD(g0);
    // (the machine scheduler otherwise sinks the last operand load and the scale load of the group behind the first waits of the query
    // copy below -- a second round trip in front of the first MFMA; tests/test_isa_guard_cpu.py)
    __builtin_amdgcn_sched_barrier(0);
#endif
    // The block's query operand (64*AFF_NQ rows x [hi|lo] x 256 B) is one contiguous run per array: copy it through LDS
    // with whole-line loads (fragment-shaped global loads touch 16 half-used lines per instruction and were most of this
    // kernel's fixed cost), then pull each wave's B fragments into registers.  Uses the A buffers before the K loop.
    {
        constexpr int QROWS = 64 * AFF_NQ, NCH = QROWS * 16 / 256;      // 16-B chunks per thread per array
        const int q0 = bx * QROWS;
        au32x4 tb[2][NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + 256 * i, row = q >> 4;
            const int jr = min(q0 + row, p.HWp - 1);                    // B rows exist up to HWp
            tb[0][i] = *reinterpret_cast<const au32x4*>(p.Bhi + (long)jr * 128 + (q & 15) * 8);
            tb[1][i] = *reinterpret_cast<const au32x4*>(p.Blo + (long)jr * 128 + (q & 15) * 8);
        }
        au32x4* lb = reinterpret_cast<au32x4*>(aff_smem);               // [hi|lo][QROWS][16 chunks], chunk ^ (row & 15)
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int q = tid + 256 * i, row = q >> 4;
            lb[row * 16 + ((q & 15) ^ (row & 15))] = tb[0][i];
            lb[QROWS * 16 + row * 16 + ((q & 15) ^ (row & 15))] = tb[1][i];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < AFF_NQ; ++u) {
            const int row = wave * (16 * AFF_NQ) + u * 16 + l15;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bh[u][ks] = __builtin_bit_cast(bf16x8, lb[row * 16 + ((ks * 4 + l4) ^ l15)]);
                bl[u][ks] = __builtin_bit_cast(bf16x8, lb[QROWS * 16 + row * 16 + ((ks * 4 + l4) ^ l15)]);
            }
        }
        __syncthreads();                                                // the A staging below overwrites this area
    }
    ATL(2)
#ifdef AFF_OLD_PROLOGUE
    AFF_LOAD(g0);
#endif
#pragma unroll
    for (int u = 0; u < AFF_NQ; ++u) {
        ncj[u] = jvalid[u] ? -craw[u] : 0.f;
        const float tcut = traw[u] - fabsf(traw[u]) * 1e-6f - 1e-30f;   // never lose the k-th element to 1 ulp
        thr[u] = (mode == 1 && jvalid[u]) ? tcut : INFINITY;
    }
#pragma unroll
    for (int u = 0; u < AFF_NQ; ++u) gcur[u] = gq[u];
    AFF_STORE(0);
    ATL(3)
    AFF_SYNC();
    int buf = 0;
    for (int gg = g0; gg < g1; gg += AFF_TG) {
        ATL(4)
        const bool more = gg + AFF_TG < g1;
        if (more) AFF_LOAD(gg + AFF_TG);
        ATL(5)
        float gm[AFF_NQ][AFF_TG];
        if constexpr (mode == 0 && AFF_NQ == 2) {
            // Pass 0 works on TWO tiles at a time: 2 tiles x 2 query sets = four independent accumulator chains of 12 dependent MFMAs each
            // (one chain is latency-bound: the timeline showed ~950 cycles per tile for 384 cycles of MFMA issue per wave, two waves per
            // SIMD).  Every chain runs its MFMAs in the order of the one-tile form (cross terms ks = 0..3, then hi * hi), so the bits are
            // those of rounds 1-4.  A tile past the block's range (only behind g1, in the last group) reads the clamped copy that AFF_LOAD
            // staged in its place; its maxima are replaced by -inf.
#pragma unroll
            for (int tp = 0; tp < AFF_TG; tp += 2) {
                ATL(6 + tp)
                bf16x8 ah[2][4], al[2][4];
                f32x4 sc[2], pd[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int row = (tp + tt) * 16 + l15;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        ah[tt][ks] = __builtin_bit_cast(bf16x8, lds[buf][row * 16 + ((ks * 4 + l4) ^ l15)]);
                        al[tt][ks] = __builtin_bit_cast(bf16x8, lds[buf][1024 + row * 16 + ((ks * 4 + l4) ^ l15)]);
                    }
                    sc[tt] = *reinterpret_cast<const f32x4*>(&lsc[buf][(tp + tt) * 16 + l4 * 4]);
                    pd[tt] = *reinterpret_cast<const f32x4*>(&lpad[buf][(tp + tt) * 16 + l4 * 4]);
                }
                f32x4 acc[2][2];
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) acc[u][tt] = (f32x4){ncj[u], ncj[u], ncj[u], ncj[u]};
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {                        // small cross terms first
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tt][ks], bl[u][ks], acc[u][tt], 0, 0, 0);
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[tt][ks], bh[u][ks], acc[u][tt], 0, 0, 0);
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                    for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) acc[u][tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[tt][ks], bh[u][ks], acc[u][tt], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        float sv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) sv[q] = fmaf(sc[tt][q], acc[u][tt][q], pd[tt][q]);      // scale_i (A.B - c_j), -inf on padding rows
                        const float m_ = rows_max(fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3])));
                        gm[u][tp + tt] = gg + tp + tt < g1 ? m_ : -INFINITY;
                    }
            }
        } else
#pragma unroll
        for (int t = 0; t < AFF_TG; ++t) {
#pragma unroll
            for (int u = 0; u < AFF_NQ; ++u) gm[u][t] = -INFINITY;
            const int g = gg + t;
            ATL(6 + t)
            if (g < g1) {                                               // block-uniform
                bool need[AFF_NQ];
                bool any = false;
#pragma unroll
                for (int u = 0; u < AFF_NQ; ++u) {
                    need[u] = !skip || __ballot(jvalid[u] && gcur[u][t] >= thr[u]) != 0;      // wave-uniform
                    any |= need[u];
                }
                if (any) {
                    int slot0, nvalid;
                    if (mode == 1) tile_slot(g, slot0, nvalid);
                    const int row = t * 16 + l15;
                    bf16x8 ah[4], al[4];
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        ah[ks] = __builtin_bit_cast(bf16x8, lds[buf][row * 16 + ((ks * 4 + l4) ^ l15)]);
                        al[ks] = __builtin_bit_cast(bf16x8, lds[buf][1024 + row * 16 + ((ks * 4 + l4) ^ l15)]);
                    }
                    // lane holds tokens l4*4 + q (q = 0..3) of the tile for its query
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(&lsc[buf][t * 16 + l4 * 4]);
                    const f32x4 pd = *reinterpret_cast<const f32x4*>(&lpad[buf][t * 16 + l4 * 4]);
#pragma unroll
                    for (int u = 0; u < AFF_NQ; ++u) {
                        if (!need[u]) continue;
                        f32x4 acc = {ncj[u], ncj[u], ncj[u], ncj[u]};
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {                // small cross terms first
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl[u][ks], acc, 0, 0, 0);
                            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh[u][ks], acc, 0, 0, 0);
                        }
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh[u][ks], acc, 0, 0, 0);
                        float s[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) s[q] = fmaf(sc[q], acc[q], pd[q]);          // scale_i (A.B - c_j), -inf on padding rows
                        const float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                        if (mode == 0) {
                            gm[u][t] = rows_max(mx);
                        } else if (__ballot(jvalid[u] && mx >= thr[u])) {   // wave-uniform: some lane has a candidate in this tile
                            AFF_APPEND_HITS(s, jq[u], thr[u], jvalid[u], slot0 + l4 * 4, wq0, 16 * AFF_NQ)
                        }
                    }
                }
            }
        }
        if (mode == 0 && l4 == 0) {                                   // 4 tile maxima per query: one 16-B store
#pragma unroll
            for (int u = 0; u < AFF_NQ; ++u)
                if (jq[u] < p.HWp)
                    *reinterpret_cast<f32x4*>(p.gmax_or_tau + (long)jq[u] * p.Gld + gg) = (f32x4){gm[u][0], gm[u][1], gm[u][2], gm[u][3]};
        }
        ATL(10)
        if (more) {
            AFF_STORE(buf ^ 1);
#pragma unroll
            for (int u = 0; u < AFF_NQ; ++u) gcur[u] = gq[u];
        }
        ATL(3)
        AFF_SYNC();
        buf ^= 1;
    }
    ATL(11)
    if (mode == 0) { ATL_DUMP(logical, nb) }
#undef AFF_LOAD
#undef AFF_STORE
    if (mode == 1) {                                                    // flush this wave's candidates: one dense burst of global atomics
        AFF_FLUSH_LIST(wcount, wq0, 16 * AFF_NQ);
#ifdef AFF_TIMELINE
        __builtin_amdgcn_s_waitcnt(0);                                  // (the stamp behind the flush counts the atomics' round trips)
        ATL(12)
        ATL(13 + (min(wcount, 240) >> 4 << 4))                          // (id 13 + 16 * (candidates of this wave / 16): how long was its list)
        ATL_DUMP(logical, nb)
#endif
    }
}

// ---- AFF_SCORE with 64 queries per wave (round 4; i[12] == 4) ---------------------------------------------------------------------
// Why: in aff_score_kernel<2, *> one set of 8 A-fragment reads (8 KB per wave and tile) feeds 24 MFMAs; with the two resident blocks of
// a CU that is 0.67 LDS-port cycles per MFMA cycle before the ds_write staging (+25 %) and the bank conflicts (0.23 of the accesses,
// profiles/r03_summary.json): the loop ran at the LDS port, not at the matrix pipe (MFMA utilisation 0.33).  Here
//   * a wave owns FOUR 16-query column sets (its B fragments: 128 VGPRs, loaded straight from global in fragment layout -- one
//     round trip, together with the first A group), so the same 8 reads feed 48 MFMAs and a 256-query block streams the memory
//     operand 7 times per frame instead of 13;
//   * the A tiles go L2 -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 4 rows per instruction, the XOR swizzle applied to the
//     per-lane SOURCE chunk as in conv_dma.hip): no staging registers, no ds_write traffic, 8 DMA instructions per wave and group
//     instead of 8 loads + 8 ds_writes per THREAD;
//   * same arithmetic per (tile, query set) as aff_score_kernel: -c_j as the accumulator's start, cross terms first, one fma per score.
typedef __amdgpu_buffer_rsrc_t aff_rsrc_t;
#define AF4_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define AF4_PRE 4096
#define AF4_STAGE (2 * 64 * 256)                      // [hi | lo] x 64 rows x 256 B
#define AF4_LDS_BYTES (2 * AF4_STAGE + AFF_LCAP * 12 + 16 + 2 * 2 * 64 * 4 + 4 * 128 * 4)

template <int NQ, int AFF_MODE, bool JT = false>      // NQ: 16-query column sets per wave (2 or 4); JT: one bank per stacked frame (see ScoreParams::tbl)
__global__ __launch_bounds__(256) void aff_score4_kernel(ScoreParams p) {
#if __HIP_DEVICE_COMPILE__
    if (p.prio) __builtin_amdgcn_s_setprio(1);
    constexpr int mode = AFF_MODE, WQ = 16 * NQ;          // WQ: queries per wave
    extern __shared__ __attribute__((aligned(16))) unsigned char aff_smem[];
    int* l_j = reinterpret_cast<int*>(aff_smem + 2 * AF4_STAGE);
    int* l_idx = l_j + AFF_LCAP;
    float* l_val = reinterpret_cast<float*>(l_idx + AFF_LCAP);
    int* l_n = reinterpret_cast<int*>(l_val + AFF_LCAP);
    float (*lsc)[64] = reinterpret_cast<float (*)[64]>(l_n + 4);        // [stage][row of the group]: scale_i (0 for padding rows)
    float (*lpad)[64] = lsc + 2;                                        // [stage][row]: 0, or -inf for padding rows
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;                                                         // XCD-aware mapping: consecutive logical blocks share the token chunk
    const int nb = gridDim.x * gridDim.y;
    int logical;
    {
        const int id = blockIdx.y * gridDim.x + blockIdx.x;
        const int xcd = id & 7, kq = id >> 3, q = nb >> 3, r = nb & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + kq;
        by = logical / (int)gridDim.x;
        bx = logical - by * (int)gridDim.x;
    }
    ATL_DECL(aff_smem + 2 * AF4_STAGE + 4096)
    ATL(0)
    const int g0 = AFF_TG * (by * p.gbase + min(by, p.grem));
    const int g1 = min(g0 + AFF_TG * (p.gbase + (by < p.grem ? 1 : 0)), p.G);
    const int T0 = (p.rn[0] + 15) >> 4, T1 = (p.nranges > 1) ? ((p.rn[1] + 15) >> 4) : 0;
    auto tile_slot = [&](int g, int& slot0, int& nvalid) {              // token slot of the first row / valid rows of tile g (wave-uniform)
        int start, n, lt;
        if (g < T0) { lt = g; start = p.rs[0]; n = p.rn[0]; }
        else if (g < T0 + T1) { lt = g - T0; start = p.rs[1]; n = p.rn[1]; }
        else { lt = g - T0 - T1; start = p.rs[2]; n = p.rn[2]; }
        slot0 = start + lt * 16;
        nvalid = min(16, n - lt * 16);
    };
    const bf16_t* Ahi_ = p.Ahi; const bf16_t* Alo_ = p.Alo; const float* scale_ = p.scale;
    if constexpr (JT) {                                                 // the bank of this block's queries (4 * WQ rows inside ONE frame)
        const int b_ = __builtin_amdgcn_readfirstlane(((bx * (4 * WQ)) / p.HWpf) % p.nbanks);
        const aff_ctbl tb_ = (aff_ctbl)p.tbl + 3 * b_;        // (constant address space: scalar loads -- a plain global load may not be scalar in a kernel that stores)
        Ahi_ = reinterpret_cast<const bf16_t*>(tb_[0]); Alo_ = reinterpret_cast<const bf16_t*>(tb_[1]); scale_ = reinterpret_cast<const float*>(tb_[2]);
    }
    const aff_rsrc_t rhi = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<bf16_t*>(JT ? Ahi_ : p.Ahi)) - AF4_PRE, 0, 0x7fffffff, 0x00020000);
    const aff_rsrc_t rlo = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<char*>(const_cast<bf16_t*>(JT ? Alo_ : p.Alo)) - AF4_PRE, 0, 0x7fffffff, 0x00020000);
    float st_sc = 0.f;
    int st_nv = 0;
    f32x4 gq[NQ];                                                       // pass-0 maxima of the NEXT group's 4 tiles (mode 1 + skip)
#pragma unroll
    for (int u = 0; u < NQ; ++u) gq[u] = (f32x4){INFINITY, INFINITY, INFINITY, INFINITY};
    const bool skip = mode == 1 && (p.mode & 2);                        // pass-0 maxima available: skip tiles without candidates
    const float* tau_p = p.gmax_or_tau;
    const float* gmax_p = p.gmax_or_tau - (long)p.HWp * p.Gld;
    const int wq0 = bx * (4 * WQ) + wave * WQ;                               // first query of this wave
    const bool wave_on = wq0 < p.HWp;                                   // (the last query block may be half empty: such waves only stage)
    int jq[NQ];
    bool jvalid[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) { jq[u] = wq0 + u * 16 + l15; jvalid[u] = jq[u] < p.HWp && jq[u] % p.HWpf < p.HW; }
    // wave w stages tile w of a group: 4 + 4 DMA pieces of 4 rows x 256 B; lane -> (row pc * 4 + (lane >> 4), chunk position lane & 15),
    // which holds source chunk (lane & 15) ^ row (the fragment reads apply the same XOR)
// s_waitcnt vmcnt(0) lgkmcnt(0) + raw s_barrier.  Explicit: hipcc does not count an LDS-DMA as something a __syncthreads() has to
// wait for (it emitted vmcnt(32) here: only the loads it knows the LDS readers depend on), see conv_dma.hip WAIT_VMCNT_LDS.
#define AF4_SYNC() { __builtin_amdgcn_s_waitcnt(0x0070); asm volatile("" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
#define AF4_PIECE(PC)                                                                                      \
        {                                                                                                  \
            const int rr_ = (PC) * 4 + l4;                                                                 \
            /* (the instruction offset PC * 1024 moves the LDS address AND the global one: taken out of the latter again; the */ \
            /* descriptors start AF4_PRE bytes in front of the banks so that the offset of a clamped row stays positive) */ \
            const unsigned vo_ = (unsigned)(min(rr_, nv_ - 1) * 256 + ((l15 ^ rr_) << 4) + (AF4_PRE - (PC) * 1024)); \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rhi, AF4_LDS_PTR(dst_), 16, vo_, so_, (PC) * 1024, 0);  \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rlo, AF4_LDS_PTR(dst_ + 16384), 16, vo_, so_, (PC) * 1024, 0); \
        }
#define AF4_LOAD(GRP, STG)                                                                                 \
    {                                                                                                      \
        int gt_ = (GRP) + wave;                                                                            \
        gt_ = gt_ < g1 ? gt_ : g1 - 1;                         /* clamp: rows of missing tiles are never used */ \
        int slot0_, nv_;                                                                                   \
        tile_slot(gt_, slot0_, nv_);                                                                       \
        const unsigned so_ = (unsigned)slot0_ * 256u;                                                      \
        unsigned char* const dst_ = aff_smem + (STG) * AF4_STAGE + wave * 4096;                            \
        /* the per-token scale: an unconditional (clamped) load whose value is looked at only where it is written to LDS, one */ \
        /* group later (a select right here made hipcc wait -- vmcnt(0) -- in the middle of the DMA pieces) */ \
        st_sc = JT ? ((aff_gf32)scale_)[slot0_ + min(l15, nv_ - 1)] : p.scale[slot0_ + min(l15, nv_ - 1)];   \
        st_nv = nv_;                                                                                       \
        AF4_PIECE(0) AF4_PIECE(1) AF4_PIECE(2) AF4_PIECE(3)                                                \
        if (skip && wave_on) {                                                                             \
            _Pragma("unroll") for (int u = 0; u < NQ; ++u)                                                 \
                gq[u] = *reinterpret_cast<const f32x4*>(gmax_p + (long)min(jq[u], p.HWp - 1) * p.Gld + (GRP)); \
        }                                                                                                  \
    }
    AF4_LOAD(g0, 0);
    ATL(1)
    // B fragments of the wave's 64 queries, straight into registers (fragment layout: lane = (query l15, k chunk ks * 4 + l4))
    bf16x8 bh[NQ][4], bl[NQ][4];
    float ncj[NQ], thr[NQ];
    {
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const long row = min(jq[u], p.HWp - 1);                     // B rows exist up to HWp
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                bh[u][ks] = __builtin_bit_cast(bf16x8, *(aff_gptr16)(p.Bhi + row * 128 + (ks * 4 + l4) * 8));
                bl[u][ks] = __builtin_bit_cast(bf16x8, *(aff_gptr16)(p.Blo + row * 128 + (ks * 4 + l4) * 8));
            }
            const float cj = p.c[min(jq[u], p.HWp - 1)];               // (c has HWp entries; unconditional loads, selects afterwards)
            ncj[u] = jvalid[u] ? -cj : 0.f;
            thr[u] = INFINITY;
            if (mode == 1) {
                const float tau = tau_p[min(jq[u], p.HWp - 1)];
                thr[u] = jvalid[u] ? tau - fabsf(tau) * 1e-6f - 1e-30f : INFINITY;     // never lose the k-th element to 1 ulp
            }
        }
    }
    ATL(2)
    int* wl_j = l_j + wave * AFF_WCAP; int* wl_idx = l_idx + wave * AFF_WCAP; float* wl_val = l_val + wave * AFF_WCAP;
    int* const wl_cnt = reinterpret_cast<int*>(lpad + 2) + wave * 128;  // (see aff_score_kernel)
    if (mode == 1) { wl_cnt[lane] = 0; }
    int wcount = 0;                                                     // wave-uniform fill of the wave's list
    f32x4 gcur[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) gcur[u] = gq[u];
    if (l4 == 0) { lsc[0][wave * 16 + l15] = l15 < st_nv ? st_sc : 0.f; lpad[0][wave * 16 + l15] = l15 < st_nv ? 0.f : -INFINITY; }
    int buf = 0;
    f32x4* const gmh = reinterpret_cast<f32x4*>(l_j);                   // [256 queries]: the maxima of the group just computed (mode 0)
    float* const gmf = reinterpret_cast<float*>(l_j);
#define AF4_FLUSH(GRP)                                                                                     \
    if (wave_on && l4 == 0) {                                                                              \
        _Pragma("unroll") for (int u = 0; u < NQ; ++u)                                                     \
            if (jq[u] < p.HWp) *reinterpret_cast<f32x4*>(p.gmax_or_tau + (long)jq[u] * p.Gld + (GRP)) = gmh[wave * WQ + u * 16 + l15]; \
    }
    for (int gg = g0; gg < g1; gg += AFF_TG) {
        ATL(3)
        AF4_SYNC();                                                     // group gg has landed (vmcnt(0)); everybody is done with the other stage
        ATL(4)
        const bool more = gg + AFF_TG < g1;
        if (more) AF4_LOAD(gg + AFF_TG, buf ^ 1);
        if (mode == 0 && gg > g0) AF4_FLUSH(gg - AFF_TG);
        ATL(5)
        const au32x4* const lA = reinterpret_cast<const au32x4*>(aff_smem + buf * AF4_STAGE);
        if (wave_on) {
#pragma unroll
            for (int t = 0; t < AFF_TG; ++t) {
                const int g = gg + t;
                ATL(6 + t)
                if (mode == 0 && g >= g1 && l4 == 0) {                  // (a tile past the block's range: -inf, as aff_score_kernel stores)
#pragma unroll
                    for (int u = 0; u < NQ; ++u) gmf[(wave * WQ + u * 16 + l15) * 4 + t] = -INFINITY;
                }
                if (g < g1) {                                           // block-uniform
                    bool need[NQ];
                    bool any = false;
#pragma unroll
                    for (int u = 0; u < NQ; ++u) {
                        need[u] = !skip || __ballot(jvalid[u] && gcur[u][t] >= thr[u]) != 0;      // wave-uniform
                        any |= need[u];
                    }
                    if (any) {
                        int slot0 = 0, nvalid = 0;
                        if (mode == 1) tile_slot(g, slot0, nvalid);
                        const int row = t * 16 + l15;
                        bf16x8 ah[4], al[4];
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            ah[ks] = __builtin_bit_cast(bf16x8, lA[row * 16 + ((ks * 4 + l4) ^ l15)]);
                            al[ks] = __builtin_bit_cast(bf16x8, lA[1024 + row * 16 + ((ks * 4 + l4) ^ l15)]);
                        }
                        // lane holds tokens l4*4 + q (q = 0..3) of the tile for its query
                        const f32x4 sc = *reinterpret_cast<const f32x4*>(&lsc[buf][t * 16 + l4 * 4]);
                        const f32x4 pd = *reinterpret_cast<const f32x4*>(&lpad[buf][t * 16 + l4 * 4]);
#pragma unroll
                        for (int u = 0; u < NQ; ++u) {
                            if (mode == 1 && !need[u]) continue;
                            f32x4 acc = {ncj[u], ncj[u], ncj[u], ncj[u]};
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) {            // small cross terms first
                                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bl[u][ks], acc, 0, 0, 0);
                                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al[ks], bh[u][ks], acc, 0, 0, 0);
                            }
#pragma unroll
                            for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[ks], bh[u][ks], acc, 0, 0, 0);
                            float s[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) s[q] = fmaf(sc[q], acc[q], pd[q]);      // scale_i (A.B - c_j), -inf on padding rows
                            const float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
                            if (mode == 0) {
                                const float m_ = rows_max(mx);
                                if (l4 == 0) gmf[(wave * WQ + u * 16 + l15) * 4 + t] = m_;      // parked in LDS until AF4_FLUSH
                            } else if (__ballot(jvalid[u] && mx >= thr[u])) {   // wave-uniform: some lane has a candidate in this tile
                                AFF_APPEND_HITS(s, jq[u], thr[u], jvalid[u], slot0 + l4 * 4, wq0, WQ)
                            }
                        }
                    }
                }
            }
            // 4 tile maxima per query = one 16-B global store -- issued one iteration LATER (AF4_FLUSH): the barrier at the top of
            // the loop waits with vmcnt(0) for this group's DMA, and a store issued here would be waited for as well (a write round trip
            // per group: measured 3.5 us per group against 1.3 us of MFMA).  The maxima wait in the (mode-1 only) candidate-list area.
        }
        ATL(10)
        if (more) {
            if (l4 == 0) { lsc[buf ^ 1][wave * 16 + l15] = l15 < st_nv ? st_sc : 0.f; lpad[buf ^ 1][wave * 16 + l15] = l15 < st_nv ? 0.f : -INFINITY; }
#pragma unroll
            for (int u = 0; u < NQ; ++u) gcur[u] = gq[u];
        }
        buf ^= 1;
    }
    if (mode == 0 && g1 > g0) {                                         // the last group's maxima (each lane reads back what it wrote)
        __builtin_amdgcn_s_waitcnt(0xc07f);                             // lgkmcnt(0)
        const int last = g0 + ((g1 - g0 - 1) / AFF_TG) * AFF_TG;
        AF4_FLUSH(last);
    }
    ATL(11)
    if (mode == 0) { ATL_DUMP(logical, nb) }
#undef AF4_FLUSH
#undef AF4_LOAD
#undef AF4_PIECE
#undef AF4_SYNC
    if (mode == 1) AFF_FLUSH_LIST(wcount, wq0, WQ);                     // flush this wave's candidates
#endif
}

__device__ __forceinline__ uint32_t f2key(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key2f(uint32_t k) {
    uint32_t u = (k & 0x80000000u) ? (k & 0x7fffffffu) : ~k;
    return __uint_as_float(u);
}

// One wave per query column, 4 waves per block: exact k-th largest tile maximum, all in registers.  The G maxima of the column sit
// MAXV per lane as order-preserving 32-bit keys; the answer is built bit by bit from the top (x |= bit while count(key >= x) >= k):
// per step one compare-and-add per register and one wave reduction -- no LDS atomics (the 4 x 8-bit histogram version below, kept for
// G > 4096, needed 11 us for 1620 columns of 667 values; this one ~3 us).
// Side jobs riding on AFF_SELECT (two fill launches and two tick launches less per read-out): the candidate counters of pass 1 are
// cleared (one int per query, stride AFF_CSTRIDE) and the life counters of up to two token ranges advance by one (USAGE_TICK).
struct SelectSide { int* count; float* lifeA; float* lifeB; int nA, nB; int zeroA; int HWpf, nrows; int prio; };    // zeroA: range A is cleared, not advanced
// HWpf / nrows: query rows per frame / in total (several frames per launch: row j is a real query iff j % HWpf < HW; one frame: nrows = HW)
__device__ __forceinline__ void select_side_jobs(const SelectSide& sd, int HW) {
    if (sd.prio) __builtin_amdgcn_s_setprio(1);
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if (sd.count) for (int q = gid; q < sd.nrows; q += nth) sd.count[q * AFF_CSTRIDE] = 0;
    if (sd.lifeA) for (int t = gid; t < sd.nA; t += nth) sd.lifeA[t] = sd.zeroA ? 0.f : sd.lifeA[t] + 1.f;
    if (sd.lifeB) for (int t = gid; t < sd.nB; t += nth) sd.lifeB[t] += 1.f;
}

template <int MAXV>
__global__ __launch_bounds__(256) void aff_select_reg_kernel(const float* __restrict__ gmax, float* __restrict__ tau,
                                                             int HW, int Gld, int G, int k, SelectSide sd) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    select_side_jobs(sd, HW);
    if (j >= sd.nrows || j % sd.HWpf >= HW) return;          // whole wave exits together
    if (G < k) { if (lane == 0) tau[j] = -INFINITY; return; }
    uint32_t key[MAXV];
    float raw[MAXV];
    const float* row = gmax + (long)j * Gld;
#pragma unroll
    for (int r = 0; r < MAXV; ++r) raw[r] = row[min(r * 64 + lane, G - 1)];      // unconditional (clamped): all loads in flight at once
#pragma unroll
    for (int r = 0; r < MAXV; ++r) key[r] = r * 64 + lane < G ? f2key(raw[r]) : 0u;  // padding sorts below every real value (-inf -> 0x007fffff)
    // (counted on the VALU, one wave reduction per bit: the ballot + s_bcnt1 + s_add form made the kernel scalar-unit-bound --
    // 1700 SALU instructions per wave on the ONE scalar unit the 16 waves of a CU share: 17 us instead of 3)
    uint32_t x = 0;
    for (int b = 31; b >= 0; --b) {
        const uint32_t t = x | (1u << b);
        int c = 0;
#pragma unroll
        for (int r = 0; r < MAXV; ++r) c += key[r] >= t ? 1 : 0;
        c = wave_sum_i32(c);                                 // DPP + permlane swaps: ~60 cycles (six dependent ds_bpermute shuffles: ~600)
        x = c >= k ? t : x;                                  // wave-uniform
    }
    if (lane == 0) tau[j] = key2f(x);
}

// one wave per query column; 4 waves per block.  Exact k-th largest by 4x8-bit radix select.
__global__ __launch_bounds__(256) void aff_select_kernel(const float* __restrict__ gmax, float* __restrict__ tau,
                                                         int HW, int Gld, int G, int k, SelectSide sd) {
    __shared__ int hist[4][256];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + wave;
    select_side_jobs(sd, HW);
    if (j >= sd.nrows || j % sd.HWpf >= HW) return;          // whole wave exits together
    if (G < k) { if (lane == 0) tau[j] = -INFINITY; return; }
    uint32_t prefix = 0, mask = 0;
    int remaining = k;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = lane; b < 256; b += 64) hist[wave][b] = 0;
        __builtin_amdgcn_wave_barrier();
        for (int g = lane; g < G; g += 64) {
            uint32_t key = f2key(gmax[(long)j * Gld + g]);
            if ((key & mask) == prefix) atomicAdd(&hist[wave][(key >> shift) & 255], 1);
        }
        __builtin_amdgcn_wave_barrier();
        int h0 = hist[wave][4 * lane], h1 = hist[wave][4 * lane + 1], h2 = hist[wave][4 * lane + 2], h3 = hist[wave][4 * lane + 3];
        int sl = h0 + h1 + h2 + h3;
        int x = sl;                                          // inclusive suffix scan over lanes (high bins = high lanes)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            int yv = __shfl_down(x, off, 64);
            if (lane + off < 64) x += yv;
        }
        int above = x - sl;
        bool mine = (above < remaining) && (remaining <= x);
        int bin = 0, rem_new = 0;
        if (mine) {
            int cum = above;
            int hh[4] = {h0, h1, h2, h3};
            bin = 4 * lane; rem_new = remaining - cum;
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                if (cum + hh[t] >= remaining) { bin = 4 * lane + t; rem_new = remaining - cum; break; }
                cum += hh[t];
            }
        }
        unsigned long long bal = __ballot(mine);
        int src = __ffsll((long long)bal) - 1;
        bin = __shfl(bin, src, 64);
        remaining = __shfl(rem_new, src, 64);
        prefix |= ((uint32_t)bin) << shift;
        mask |= 0xffu << shift;
        __builtin_amdgcn_wave_barrier();
    }
    if (lane == 0) tau[j] = key2f(prefix);
}

#define RO_THREADS 128
typedef __attribute__((ext_vector_type(4))) unsigned int ro_u32x4;
typedef __attribute__((ext_vector_type(4))) int ro_i32x4;
// (a pointer loaded from memory has no known address space: hipcc would emit flat_load, whose lgkmcnt accounting couples the
// gather with every LDS read of the selection lists)
typedef const __attribute__((address_space(1))) ro_u32x4* ro_gptr;
#define RO_MAXK 64
#define RO_PRUNE 64                                    // lists longer than this are cut to the entries >= the top_k-th value before ranking
// one block per query column
__global__ __launch_bounds__(RO_THREADS) void aff_readout_kernel(const float* __restrict__ cand_val, const int* __restrict__ cand_idx,
                                                                 const int* __restrict__ count, const uint64_t* __restrict__ vptrs,
                                                                 float* __restrict__ usage, bf16_t* __restrict__ y, int* __restrict__ overflow,
                                                                 int HW, int cap, int topk, int K, int CV, int HWpf, int nrows, int ustride, int prio, int nbanks, int ufx) {
    if (prio) __builtin_amdgcn_s_setprio(1);
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    float* cv = reinterpret_cast<float*>(lds_raw);             // [cap]
    int* ci = reinterpret_cast<int*>(lds_raw + (size_t)cap * 4);  // [cap]
    __shared__ __attribute__((aligned(16))) float sel_v[RO_MAXK];
    __shared__ __attribute__((aligned(16))) int sel_i[RO_MAXK];
    __shared__ __attribute__((aligned(16))) float sel_w[RO_MAXK];
    __shared__ uint32_t prune_key;
    __shared__ int prune_n;
    // XCD-aware: hardware block b runs on XCD b % 8; give each XCD one contiguous stripe of queries (neighbouring pixels
    // select overlapping memory tokens, so a stripe's value rows stay in that XCD's L2 instead of all 8 L2s fetching all)
    // Several frames per launch (nrows = frames x HWpf query rows, row j real iff j % HWpf < HW): frame f's read-out goes to
    // y + f * K * HW * CV and its usage to usage + f * ustride (per-frame side buffers: committed frame by frame, memory_manager.py).
    const int per = (nrows + 7) >> 3;
    const int j = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int tid = threadIdx.x;
    if (j >= nrows || (int)(blockIdx.x >> 3) >= per) return;
    const int fr = j / HWpf, jl = j - fr * HWpf;
    if (jl >= HW) return;
    y += (long)fr * K * HW * CV;
    if (usage) usage += (long)fr * ustride * (ufx ? 2 : 1);     // (fixed-point usage: 8-byte counters)
    if (nbanks > 1) vptrs += (long)(fr % nbanks) * K;          // (ABI 4) clips in lock step: frame fr reads bank fr % nbanks, whose K value-bank bases follow each other
    // every global round trip that does not depend on another is issued up front: the count, the first RO_THREADS
    // candidates (typical fill is 30-50 of the 1024 slots), the bank base pointer of this thread's object
    const int C8 = CV >> 3;
    const int o_mine = min(tid / C8, K - 1);
    const bf16_t* Vbase = reinterpret_cast<const bf16_t*>(vptrs[o_mine]);
    const int t0 = tid < cap ? tid : cap - 1;
    const float v0 = cand_val[(long)j * cap + t0];
    const int i0 = cand_idx[(long)j * cap + t0];
    const int first = cand_idx[(long)j * cap];
    int cnt = count[j * AFF_CSTRIDE];
    if (cnt > cap) { if (tid == 0) atomicAdd(overflow, 1); cnt = cap; }
    const int cnt4 = (cnt + 3) & ~3;                           // LDS lists padded to a multiple of 4 with (-inf, INT_MAX)
    if (tid < cnt) { cv[tid] = v0; ci[tid] = i0; }
    for (int t = tid + RO_THREADS; t < cnt; t += RO_THREADS) { cv[t] = cand_val[(long)j * cap + t]; ci[t] = cand_idx[(long)j * cap + t]; }
    if (tid < cnt4 - cnt) { cv[cnt + tid] = -INFINITY; ci[cnt + tid] = 0x7fffffff; }
    const int nsel = min(cnt, topk);
    // padding / unassigned ranks: any valid slot (gathered with weight 0)
    if (tid < RO_MAXK) { sel_v[tid] = -INFINITY; sel_i[tid] = cnt > 0 ? first : 0; sel_w[tid] = 0.f; }
    __syncthreads();
    // Long lists are cut first.  The threshold of pass 1 is the top_k-th largest TILE maximum, so a query whose best tiles are
    // homogeneous brings hundreds of candidates (measured on the bench clip: median 32, 1 % of the queries ~270) and the all-pairs
    // ranking below is quadratic: those few blocks were the kernel (49 us; 17 us with every list cut to 64, tools/aff_ab.py).
    // Wave 0 finds the exact top_k-th largest VALUE of the list (bitwise construction on order-preserving keys, as aff_select_reg_kernel),
    // the block keeps the entries >= it (>= top_k of them, all ties included), and the ranking -- value descending, ties -> lower slot --
    // runs on what is left: the same top_k in the same order.
    int ncand = cnt, ncand4 = cnt4;
    if (cnt > RO_PRUNE && cnt <= 1024) {                       // block-uniform
        if (tid < 64) {
            uint32_t key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) key[r] = r * 64 + tid < cnt ? f2key(cv[min(r * 64 + tid, cnt - 1)]) : 0u;   // padding sorts below -inf
            uint32_t x = 0;
            for (int b = 31; b >= 0; --b) {
                const uint32_t t = x | (1u << b);
                int c = 0;
#pragma unroll
                for (int r = 0; r < 16; ++r) c += key[r] >= t ? 1 : 0;
                c = wave_sum_i32(c);
                x = c >= nsel ? t : x;                         // wave-uniform
            }
            if (tid == 0) { prune_key = x; prune_n = 0; }
        }
        __syncthreads();
        const uint32_t x = prune_key;
        float mv[8]; int mi[8]; bool keep[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int t = tid + r * RO_THREADS, tc = min(t, cnt - 1);
            mv[r] = cv[tc]; mi[r] = ci[tc];
            keep[r] = t < cnt && f2key(mv[r]) >= x;
        }
        __syncthreads();                                       // everybody holds its entries: the list is rebuilt in place
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (keep[r]) { const int pos = atomicAdd(&prune_n, 1); cv[pos] = mv[r]; ci[pos] = mi[r]; }     // (order is irrelevant: ranked below)
        __syncthreads();
        ncand = prune_n;
        ncand4 = (ncand + 3) & ~3;
        if (tid < ncand4 - ncand) { cv[ncand + tid] = -INFINITY; ci[ncand + tid] = 0x7fffffff; }
        __syncthreads();
    }
    // exact rank of every candidate (descending value, ties -> lower slot); the lists are read 4 entries per ds_read_b128
    // (bitwise | and &: the short-circuit forms compiled to a branch per entry)
    for (int t = tid; t < ncand; t += RO_THREADS) {
        const float v = cv[t]; const int id = ci[t];
        int rank = 0;
#pragma unroll 2
        for (int u = 0; u < ncand4; u += 4) {
            const f32x4 w = *reinterpret_cast<const f32x4*>(cv + u);
            const ro_i32x4 d = *reinterpret_cast<const ro_i32x4*>(ci + u);
#pragma unroll
            for (int e = 0; e < 4; ++e) rank += (int)(w[e] > v) | ((int)(w[e] == v) & (int)(d[e] < id));
        }
        if (rank < nsel) { sel_v[rank] = v; sel_i[rank] = id; }
    }
    __syncthreads();
    if (tid < 64) {                                            // softmax over the selected scores (wave 0)
        float e = (tid < nsel && sel_v[tid] > -INFINITY) ? expf(sel_v[tid] - sel_v[0]) : 0.f;   // unassigned rank (duplicate entries): weight 0
        float sum = wave_sum(e);
        if (tid < nsel) sel_w[tid] = e / sum;
    }
    __syncthreads();
    // sparse V gather: 16 independent 16-B row loads in flight per thread (the gather is latency-bound: ~46 KB of 512-B
    // rows per query from banks that do not fit one XCD's L2); padding entries carry weight 0 and a valid slot
    constexpr int RPR = 16;                                    // rows in flight per thread (32 would need 194 VGPRs: 4 instead of 10 resident blocks per CU)
    const int nround = (nsel + RPR - 1) / RPR;
    for (int u = tid; u < K * C8; u += RO_THREADS) {
        int o = u / C8, c8 = u - o * C8;
        const bf16_t* V = (u == tid ? Vbase : reinterpret_cast<const bf16_t*>(vptrs[o])) + c8 * 8;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int r = 0; r < nround; ++r) {
            ro_u32x4 v[RPR];
#pragma unroll
            for (int t4 = 0; t4 < RPR / 4; ++t4) {
                const ro_i32x4 si = *reinterpret_cast<const ro_i32x4*>(sel_i + r * RPR + t4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[t4 * 4 + e] = *(ro_gptr)(V + (long)si[e] * CV);   // bank pointers come from memory: force global_load
            }
            float wv[RPR];
#pragma unroll
            for (int t4 = 0; t4 < RPR / 4; ++t4) {
                const f32x4 w4 = *reinterpret_cast<const f32x4*>(sel_w + r * RPR + t4 * 4);
                wv[t4 * 4] = w4[0]; wv[t4 * 4 + 1] = w4[1]; wv[t4 * 4 + 2] = w4[2]; wv[t4 * 4 + 3] = w4[3];
            }
#pragma unroll
            for (int t = 0; t < RPR; ++t) {
                const float w = wv[t];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[2 * i] += w * __uint_as_float(v[t][i] << 16);
                    acc[2 * i + 1] += w * __uint_as_float(v[t][i] & 0xffff0000u);
                }
            }
        }
        *reinterpret_cast<uint4*>(y + ((long)o * HW + jl) * CV + c8 * 8) =
            make_uint4(pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3]), pack_bf2(acc[4], acc[5]), pack_bf2(acc[6], acc[7]));
    }
    // usage += softmax weight of every selected token (kv_memory_store.py:151-162).  Last on purpose: on gfx950 an atomic counts in
    // vmcnt like a load, so issued before the gather every wait for a value row also waited for the atomics, which serialise in
    // the L2 on the popular tokens.
    // ufx (what the product uses, round 6): the counters are unsigned 64-bit FIXED POINT (2^-40): integer atomics commute, so the sums do not
    // depend on the order in which the blocks arrive.  The float form's last bits did -- and with them, a few hundred frames later, a near-tie
    // of the long-term consolidation's usage ranking: a clip advanced in lock step left its own InferenceCore run (tools/lockstep_soak.py).
    if (usage && tid < nsel) {
        if (ufx) atomicAdd(reinterpret_cast<unsigned long long*>(usage) + sel_i[tid], (unsigned long long)(sel_w[tid] * 1099511627776.f));
        else atomicAdd(&usage[sel_i[tid]], sel_w[tid]);
    }
}

int launch_affinity(const cutie_op* op, hipStream_t s) {
    const int32_t* i = op->i;
    const uint64_t* p = op->p;
    switch (op->kind) {
        case CUTIE_OP_KEY_PREP: {
            long n = (long)i[0] * 16;
            hipLaunchKernelGGL(key_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)p[0], (const float*)p[1],
                               (bf16_t*)p[2], (bf16_t*)p[3], (float*)p[4], i[0], op->flags & 3);
            break;
        }
        case CUTIE_OP_AFF_SCORE: {
            ScoreParams sp;
            sp.Ahi = (const bf16_t*)p[0]; sp.Alo = (const bf16_t*)p[1]; sp.scale = (const float*)p[2];
            sp.Bhi = (const bf16_t*)p[3]; sp.Blo = (const bf16_t*)p[4]; sp.c = (const float*)p[5];
            sp.gmax_or_tau = (float*)p[6]; sp.cand_val = (float*)p[7]; sp.cand_idx = (int*)p[8]; sp.count = (int*)p[9];
            sp.HW = i[0]; sp.HWp = i[1]; sp.nranges = i[2];
            for (int r = 0; r < 3; ++r) { sp.rs[r] = i[3 + 2 * r]; sp.rn[r] = (r < sp.nranges) ? i[4 + 2 * r] : 0; }
            sp.G = i[9]; sp.cap = i[10]; sp.mode = i[11]; sp.Gld = (sp.G + 63) / 64 * 64;
            sp.prio = (op->flags & 64) ? 1 : 0;
            sp.HWpf = i[16] > 0 ? i[16] : sp.HWp;              // i[16]: query rows per frame when one launch serves several frames (HWp = frames x i[16])
            const bool jt = (op->flags & 4) != 0;               // (ABI 4) the stacked frames read different banks: p11 = (A_hi, A_lo, scale) per bank, i17 = banks
            sp.tbl = jt ? (const unsigned long long*)p[11] : nullptr; sp.nbanks = jt ? i[17] : 1;
            if (sp.HWp % sp.HWpf || sp.HW > sp.HWpf) { cutie_set_error("aff_score: HWp=%d is not a multiple of the rows per frame %d (HW=%d)", sp.HWp, sp.HWpf, sp.HW); return -2; }
            int G = 0;
            for (int r = 0; r < 3; ++r) G += (sp.rn[r] + 15) / 16;
            if (G != sp.G || (sp.HWp & 63) || sp.nranges < 1 || sp.nranges > 3 || (sp.Gld & 3)) { cutie_set_error("aff_score: bad ranges (G=%d vs %d, HWp=%d)", G, sp.G, sp.HWp); return -2; }
            const bool dma = i[12] == 4 || i[15] == 1;          // aff_score4_kernel (LDS-DMA staging): always for 4 sets per wave, for 2 when i[15] = 1
            const int nq = i[12] == 1 ? 1 : (i[12] == 4 ? 4 : 2);
            if (jt && (nq != 2 || !sp.tbl || sp.nbanks < 2 || (sp.HWp / sp.HWpf) % sp.nbanks || sp.HWpf % 128)) {
                cutie_set_error("aff_score (flags&4): needs 2 query sets per wave, the bank table, frames %% banks == 0 and rows per frame %% 128 == 0 (nq=%d banks=%d frames=%d rows=%d)",
                                nq, sp.nbanks, sp.HWp / sp.HWpf, sp.HWpf);
                return -2;
            }
            int qb = (sp.HWp + 64 * nq - 1) / (64 * nq);
            if (sp.mode != 0 && sp.mode != 1) { cutie_set_error("aff_score: mode %d", sp.mode); return -2; }
            const int pass = sp.mode;
            if (pass == 1 && (op->flags & 1)) sp.mode |= 2;     // pass-0 maxima precede tau in memory: tiles without candidates are skipped
            static bool lds_attr_set = false;
            if (!lds_attr_set) {
                const void* ks[12] = {reinterpret_cast<const void*>(aff_score_kernel<1, 0>), reinterpret_cast<const void*>(aff_score_kernel<1, 1>),
                                      reinterpret_cast<const void*>(aff_score_kernel<2, 0>), reinterpret_cast<const void*>(aff_score_kernel<2, 1>),
                                      reinterpret_cast<const void*>(aff_score4_kernel<4, 0>), reinterpret_cast<const void*>(aff_score4_kernel<4, 1>),
                                      reinterpret_cast<const void*>(aff_score4_kernel<2, 0>), reinterpret_cast<const void*>(aff_score4_kernel<2, 1>),
                                      reinterpret_cast<const void*>(aff_score_kernel<2, 0, true>), reinterpret_cast<const void*>(aff_score_kernel<2, 1, true>),
                                      reinterpret_cast<const void*>(aff_score4_kernel<2, 0, true>), reinterpret_cast<const void*>(aff_score4_kernel<2, 1, true>)};
                static_assert(AF4_LDS_BYTES == AFF_LDS_BYTES, "one dynamic-LDS size for all score kernels");
                for (const void* k : ks)
                    if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                        cutie_set_error("aff_score: cannot raise the dynamic LDS limit to %d bytes", AFF_LDS_BYTES);
                        return -2;
                    }
                lds_attr_set = true;
            }
            // ONE round of blocks: at most `resident` of them (2 per CU), the 4-tile groups of the memory operand split evenly over the
            // block rows (gbase or gbase + 1 groups each).  Rounds 1-4 sized a block as ceil(G * qb / resident) tiles and let the grid
            // follow: 520 blocks on 512 slots whenever the last row was not short -- the eight late blocks doubled the launch (stacked
            // frames: 160 us for 4 x 29 us of work, tools/r5_call2.sh).  At least 8 tiles per block: the block prologue (its query
            // operand, 64 KB through LDS) is amortised over the tiles.
            int resident = (nq == 4 && pass == 1) ? 256 : 512;    // (aff_score4_kernel<1> needs > 256 registers: one block per CU)
            if (i[14] > 0) resident = 256;                              // (extra LDS requested: one block per CU)
            const int NG = (G + AFF_TG - 1) / AFF_TG;
            int ny = resident / qb;
            if (ny > NG / 2) ny = NG / 2;
            if (ny < 1) ny = 1;
            if (i[13] > 0) ny = (NG + (i[13] + AFF_TG - 1) / AFF_TG - 1) / ((i[13] + AFF_TG - 1) / AFF_TG);   // (tuning override: tiles per block)
            sp.gbase = NG / ny; sp.grem = NG % ny;
            sp.tiles_per_block = AFF_TG * (sp.gbase + (sp.grem ? 1 : 0));
            const dim3 grid(qb, ny);
            const int lds4 = AFF_LDS_BYTES + (i[14] > 0 && i[14] <= 80 ? i[14] * 1024 : 0);       // (diagnostic: extra dynamic LDS = fewer resident blocks per CU)
#ifdef AFF_TIMELINE
            sp.tl = (unsigned long long*)p[10];
            const int lds2 = AFF_LDS_BYTES + (pass == 1 ? 4 * ATL_MAX * 8 : 0);      // (pass 1 parks its stamps behind the regular LDS: 3040 B, still two blocks per CU)
#else
            const int lds2 = AFF_LDS_BYTES;
#endif
            if (jt && dma && pass == 0) hipLaunchKernelGGL((aff_score4_kernel<2, 0, true>), grid, dim3(256), lds4, s, sp);
            else if (jt && dma) hipLaunchKernelGGL((aff_score4_kernel<2, 1, true>), grid, dim3(256), lds4, s, sp);
            else if (jt && pass == 0) hipLaunchKernelGGL((aff_score_kernel<2, 0, true>), grid, dim3(256), lds2, s, sp);
            else if (jt) hipLaunchKernelGGL((aff_score_kernel<2, 1, true>), grid, dim3(256), lds2, s, sp);
            else if (nq == 4 && pass == 0) hipLaunchKernelGGL((aff_score4_kernel<4, 0>), grid, dim3(256), lds4, s, sp);
            else if (nq == 4) hipLaunchKernelGGL((aff_score4_kernel<4, 1>), grid, dim3(256), lds4, s, sp);
            else if (dma && nq == 2 && pass == 0) hipLaunchKernelGGL((aff_score4_kernel<2, 0>), grid, dim3(256), lds4, s, sp);
            else if (dma && nq == 2) hipLaunchKernelGGL((aff_score4_kernel<2, 1>), grid, dim3(256), lds4, s, sp);
            else if (nq == 1 && pass == 0) hipLaunchKernelGGL((aff_score_kernel<1, 0>), grid, dim3(256), lds2, s, sp);
            else if (nq == 1) hipLaunchKernelGGL((aff_score_kernel<1, 1>), grid, dim3(256), lds2, s, sp);
            else if (pass == 0) hipLaunchKernelGGL((aff_score_kernel<2, 0>), grid, dim3(256), lds2, s, sp);
            else hipLaunchKernelGGL((aff_score_kernel<2, 1>), grid, dim3(256), lds2, s, sp);
            break;
        }
        case CUTIE_OP_AFF_SELECT: {
            // i[6] > 1: that many frames of i[1] query rows each (gmax / tau / the counters are indexed by row; i[0] real queries per frame)
            const int frames = i[6] > 1 ? i[6] : 1;
            const int nrows = frames > 1 ? frames * i[1] : i[0];
            SelectSide sd = {(int*)p[2], (float*)p[3], (float*)p[4], i[4], i[5], op->flags & 1, frames > 1 ? i[1] : nrows, nrows, (op->flags & 64) ? 1 : 0};
            const dim3 grid((nrows + 3) / 4), block(256);
            const int Gld = (i[2] + 63) / 64 * 64;
            // values per lane: the kernel counts MAXV compares per lane and bit, so MAXV follows the number of tiles in steps of 4 x 64 tiles
            // (16 | 32 | 64 only: 10.7 us at 700 tiles, 16.5 us from 1025 tiles on -- the bench clip crosses that line at 16.4 k tokens);
            // flags&2: the three coarse sizes (A/B switch).  Exact selection either way.
            const int need = (op->flags & 2) ? (i[2] <= 1024 ? 16 : i[2] <= 2048 ? 32 : 64) : (i[2] + 63) / 64;
#define AFF_SEL(MV) hipLaunchKernelGGL(aff_select_reg_kernel<MV>, grid, block, 0, s, (const float*)p[0], (float*)p[1], i[0], Gld, i[2], i[3], sd)
            if (i[2] > 4096) hipLaunchKernelGGL(aff_select_kernel, grid, block, 0, s, (const float*)p[0], (float*)p[1], i[0], Gld, i[2], i[3], sd);
            else if (need <= 4) AFF_SEL(4); else if (need <= 8) AFF_SEL(8); else if (need <= 12) AFF_SEL(12); else if (need <= 16) AFF_SEL(16);
            else if (need <= 20) AFF_SEL(20); else if (need <= 24) AFF_SEL(24); else if (need <= 28) AFF_SEL(28); else if (need <= 32) AFF_SEL(32);
            else if (need <= 40) AFF_SEL(40); else if (need <= 48) AFF_SEL(48); else if (need <= 56) AFF_SEL(56); else AFF_SEL(64);
#undef AFF_SEL
            break;
        }
        case CUTIE_OP_AFF_READOUT: {
            if (i[2] > RO_MAXK || (i[4] & 7) || (i[1] & 3)) { cutie_set_error("aff_readout: top_k <= %d, CV %% 8, cap %% 4", RO_MAXK); return -2; }
            size_t lds = (size_t)i[1] * 8;
            // i[5] > 1: that many frames of i[6] query rows each; i[7] = floats between the frames' usage buffers
            const int frames = i[5] > 1 ? i[5] : 1;
            const int nrows = frames > 1 ? frames * i[6] : i[0];
            if (i[8] > 1 && frames % i[8]) { cutie_set_error("aff_readout: %d frames over %d banks", frames, i[8]); return -2; }
            hipLaunchKernelGGL(aff_readout_kernel, dim3(((nrows + 7) / 8) * 8), dim3(RO_THREADS), lds, s, (const float*)p[0], (const int*)p[1], (const int*)p[2],
                               (const uint64_t*)p[3], (float*)p[4], (bf16_t*)p[5], (int*)p[6], i[0], i[1], i[2], i[3], i[4],
                               frames > 1 ? i[6] : nrows, nrows, i[7], (op->flags & 64) ? 1 : 0, i[8] > 1 ? i[8] : 1, (op->flags & 1) ? 1 : 0);
            break;
        }
        default:
            cutie_set_error("affinity: unknown op kind %d", op->kind);
            return -3;
    }
    return (int)hipGetLastError();
}
