// Stable segmented key / position sort for gfx950.
// Replaces segmented_sort_pairs and its kernels (src/device/sort_ops_kernel.cu:10-505, src/device/utils/sorting_radix_sort.h,
// key transforms src/device/utils/sorting_common.h:23-260) as they are used by sort_stable_kernel
// (sort_ops_kernel.cu:556-618): keys of one dtype in nseg contiguous segments of n, values = the int64 position of each
// key inside its segment, stable, ascending or descending.
//
// Integer / byte work, bit-exact. Keys are mapped to unsigned integers of the same order (floats: flip the sign bit of
// non-negatives, all bits of negatives, so -0.0 < +0.0 and NaNs sort by bit pattern, as the reference's KeyTraits do);
// a descending sort is the ascending sort of the complemented key, which keeps equal keys in input order exactly like
// the reference's reversed digit bins.
//
//   * n <= 512 (round 5): bitonic network over (key, position) composites held in REGISTERS - positions are unique, so the total order IS the
//     stable order and no ranking pass is needed. n <= 64: a wave's 64 lanes are 64 / P segments of P slots, four rows per wave in flight;
//     64 < n <= 512: one segment per wave, P / 64 slots per lane (partner distances of 64 and more pair two registers of one lane).
//     Partners come through DPP / ds_swizzle / ds_bpermute, the compare becomes the select mask by one scalar XOR with a constant of the stage;
//     no LDS memory, no barrier. [1 Mi, 64] f32: 0.62 -> 0.19 ms = 5.6 TB/s of key + position traffic (0.70 of HBM); 64 Mi keys in segments of
//     128 / 256 / 512: 0.83 / 1.01 / 1.23 ms with the LDS network this replaced.
//   * 512 < n <= 8192 (round 5): ONE block sorts ONE segment by a least-significant-digit radix sort that never leaves LDS - 8-bit digits, one
//     pass per key byte, the ballot ranking of the long-segment path below, positions carried as 16-bit values. The bitonic network needs
//     log2(P) (log2(P) + 1) / 2 = 78 LDS passes over a 4096-key segment (40 k cycles of LDS pipe); four radix passes need ~10 k cycles of VALU.
//     Padding slots hold the largest key and the largest positions: a stable sort leaves them behind every real key.
//   * n > 8192: least-significant-digit radix sort, 8-bit digits (one pass per key byte), four launches per pass:
//     tile histograms -> two-level exclusive scan (inside chunks of 64 tiles, then over the chunks) -> stable scatter. A tile is 8192 keys of up to
//     four bytes (512 threads) or 4096 eight-byte keys (256 threads), 1024 consecutive keys per wave.
//     Ranking inside a wave is by digit match: 8 ballots give each lane the set of lanes holding its digit (one three-input
//     bit operation per half and bit), the lane's rank is a popcount below itself - no per-key atomics, stable by construction.
//     The tile is first reordered in LDS, then written out, so every digit's run leaves as consecutive addresses; every XCD
//     works on one contiguous eighth of the tiles, so neighbouring tiles' runs meet in one L2. The first pass reads the caller's
//     typed keys, the last writes typed keys and int64 positions; the passes in between ping-pong through the workspace.
//     Round 5, [1, 64 Mi] f32: 2.19 -> 1.24 ms (scatter 466 -> 231 us per pass = 16 B per key at 4.6 TB/s, histogram 110 -> 65, scans 22 -> 15).
#include "common.h"

namespace kf {

enum { K_UNSIGNED = 0, K_SIGNED = 1, K_FLOAT = 2 };

template <typename U, int W> struct KeyBits {
    static constexpr U all = W == (int)sizeof(U) ? ~(U)0 : (((U)1 << (8 * (W % (int)sizeof(U)))) - 1);
    static constexpr U sign = (U)1 << (8 * W - 1);
};

template <typename U, int W> __device__ __forceinline__ U load_raw(const void *p, int64_t i) {
    if constexpr (W == 1) return ((const uint8_t *)p)[i];
    else if constexpr (W == 2) return ((const uint16_t *)p)[i];
    else if constexpr (W == 4) return ((const uint32_t *)p)[i];
    else return ((const uint64_t *)p)[i];
}
template <typename U, int W> __device__ __forceinline__ void store_raw(void *p, int64_t i, U v) {
    if constexpr (W == 1) ((uint8_t *)p)[i] = (uint8_t)v;
    else if constexpr (W == 2) ((uint16_t *)p)[i] = (uint16_t)v;
    else if constexpr (W == 4) ((uint32_t *)p)[i] = (uint32_t)v;
    else ((uint64_t *)p)[i] = (uint64_t)v;
}
// sorting_common.h:40-55 (float), 158-170 (int32), 186-202 (double), 204-240 (half / bfloat16)
template <typename U, int W, int KIND> __device__ __forceinline__ U to_ordered(U raw, U flip) {
    using B = KeyBits<U, W>;
    U o = raw;
    if constexpr (KIND == K_SIGNED) o = raw ^ B::sign;
    if constexpr (KIND == K_FLOAT) o = raw ^ ((raw & B::sign) ? B::all : B::sign);
    return o ^ flip;
}
template <typename U, int W, int KIND> __device__ __forceinline__ U from_ordered(U ord, U flip) {
    using B = KeyBits<U, W>;
    U o = ord ^ flip;
    if constexpr (KIND == K_SIGNED) o ^= B::sign;
    if constexpr (KIND == K_FLOAT) o ^= (o & B::sign) ? B::sign : B::all;
    return o;
}

struct SmallArgs {
    const void *in;
    void *out;
    int64_t *pos;
    int64_t nseg;
    int n, logp, npass, desc; // npass: radix passes to run (the key bytes that can differ between keys; all of them unless the caller knows better)
};

// ------------------------------------------------------------------------------------------
// segments of up to 64 keys: bitonic network in registers, one 64-slot row per wave instruction stream
// ------------------------------------------------------------------------------------------
// A row is 64 / P segments of P = 2^LOGP slots; slot = lane. The exchange partner lane ^ j comes through DPP (j = 1, 2, 8), ds_swizzle (4, 16) or
// ds_bpermute (32) - no LDS memory, no barrier. Which of the pair a lane keeps is a lane mask: (partner < mine) XNOR (this lane wants the
// smaller one), the second a compile-time constant of the stage - one scalar instruction between the compare and the two selects.
// A wave carries WR independent rows through the network together (their chains interleave in the instruction stream).
__device__ __forceinline__ constexpr uint64_t lanes_with_bit_clear(int b) {
    return b == 1 ? 0x5555555555555555ull : b == 2 ? 0x3333333333333333ull : b == 4 ? 0x0F0F0F0F0F0F0F0Full : b == 8 ? 0x00FF00FF00FF00FFull
         : b == 16 ? 0x0000FFFF0000FFFFull : b == 32 ? 0x00000000FFFFFFFFull : ~0ull;
}
template <int J> __device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    if constexpr (J == 1) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);  // quad_perm [2,3,0,1]
    else if constexpr (J == 8) return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true); // row_ror:8
    else if constexpr (J == 4) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x101F);                   // and 0x1f, xor 4
    else if constexpr (J == 16) return (uint32_t)__builtin_amdgcn_ds_swizzle((int)v, 0x401F);                 // and 0x1f, xor 16
    else return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(((threadIdx.x & 63) ^ 32) << 2), (int)v);
}
__device__ __forceinline__ uint32_t pick(uint64_t take, uint32_t mine, uint32_t other) {
    uint32_t r;
    asm("v_cndmask_b32 %0, %1, %2, %3" : "=v"(r) : "v"(mine), "v"(other), "s"(take));
    return r;
}

template <typename U> struct WaveSlot;
template <> struct WaveSlot<uint32_t> {
    uint32_t k, p;
    // UP: the lanes whose pair is to come out ascending (a constant of the stage)
    template <int J, uint64_t UP> __device__ __forceinline__ void step() {
        const uint32_t ok = lane_xor<J>(k), op = lane_xor<J>(p);
        const uint64_t lt = __ballot((((uint64_t)ok << 32) | op) < (((uint64_t)k << 32) | p));
        const uint64_t take = lt ^ (lanes_with_bit_clear(J) ^ UP); // = NOT (lt XOR (this lane keeps the smaller one))
        k = pick(take, k, ok);
        p = pick(take, p, op);
    }
    // the pair (lo, hi) of ONE lane: ascending (UP) or descending
    template <bool UP> static __device__ __forceinline__ void order(WaveSlot &lo, WaveSlot &hi) {
        const uint64_t lt = __ballot((((uint64_t)hi.k << 32) | hi.p) < (((uint64_t)lo.k << 32) | lo.p));
        const uint64_t take = UP ? lt : ~lt;
        const uint32_t k0 = pick(take, lo.k, hi.k), p0 = pick(take, lo.p, hi.p);
        hi.k = pick(take, hi.k, lo.k);
        hi.p = pick(take, hi.p, lo.p);
        lo.k = k0;
        lo.p = p0;
    }
};
template <> struct WaveSlot<uint64_t> {
    uint64_t k;
    uint32_t p;
    template <int J, uint64_t UP> __device__ __forceinline__ void step() {
        const uint32_t klo = (uint32_t)k, khi = (uint32_t)(k >> 32);
        const uint32_t olo = lane_xor<J>(klo), ohi = lane_xor<J>(khi), op = lane_xor<J>(p);
        const uint64_t ok = ((uint64_t)ohi << 32) | olo;
        const uint64_t lt = __ballot(ok < k || (ok == k && op < p));
        const uint64_t take = lt ^ (lanes_with_bit_clear(J) ^ UP);
        k = ((uint64_t)pick(take, khi, ohi) << 32) | pick(take, klo, olo);
        p = pick(take, p, op);
    }
    template <bool UP> static __device__ __forceinline__ void order(WaveSlot &lo, WaveSlot &hi) {
        const uint64_t lt = __ballot(hi.k < lo.k || (hi.k == lo.k && hi.p < lo.p));
        const uint64_t take = UP ? lt : ~lt;
        const uint32_t a0 = pick(take, (uint32_t)lo.k, (uint32_t)hi.k), a1 = pick(take, (uint32_t)(lo.k >> 32), (uint32_t)(hi.k >> 32)), p0 = pick(take, lo.p, hi.p);
        const uint32_t b0 = pick(take, (uint32_t)hi.k, (uint32_t)lo.k), b1 = pick(take, (uint32_t)(hi.k >> 32), (uint32_t)(lo.k >> 32));
        hi.p = pick(take, hi.p, lo.p);
        hi.k = ((uint64_t)b1 << 32) | b0;
        lo.k = ((uint64_t)a1 << 32) | a0;
        lo.p = p0;
    }
};

// The network as templates (every stage's partner distance, direction mask and register pairing is a compile-time constant):
// slot = e 64 + lane, E slots per lane, P = 64 E slots per segment (E = 1: 64 / P segments of P slots per row).
template <typename U, int P, int E, int K, int J, int e> __device__ __forceinline__ void net_slot(WaveSlot<U> (&x)[E]) {
    constexpr bool up_e = ((e * 64) & K) == 0 || K >= P * (E == 1 ? 64 / P : 1); // (the last merge is ascending in every segment: its direction bit lies outside the slot index)
    if constexpr (J >= 64) {
        if constexpr ((e & (J >> 6)) == 0) WaveSlot<U>::template order<up_e>(x[e], x[e | (J >> 6)]);
    } else {
        constexpr uint64_t UP = (K < 64 && K < P) ? lanes_with_bit_clear(K) : (up_e ? ~0ull : 0ull);
        x[e].template step<J, UP>();
    }
    if constexpr (e + 1 < E) net_slot<U, P, E, K, J, e + 1>(x);
}
template <typename U, int P, int E, int WR, int K, int J> __device__ __forceinline__ void net_from(WaveSlot<U> (&x)[WR][E]) { // stage (K, J) and all behind it
#pragma unroll
    for (int r = 0; r < WR; ++r) net_slot<U, P, E, K, J, 0>(x[r]);
    if constexpr (J > 1) net_from<U, P, E, WR, K, J / 2>(x);
    else if constexpr (K < P) net_from<U, P, E, WR, 2 * K, K>(x);
}

constexpr int kWaveRows = 4;
template <typename U, int W, int KIND, int LOGP>
__global__ __launch_bounds__(256) void sort_wave_kernel(const SmallArgs a) {
    constexpr int P = 1 << LOGP, SPW = 64 / P, WR = kWaveRows;
    const int lane = threadIdx.x & 63, il = lane & (P - 1);
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * WR;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    WaveSlot<U> x[WR][1];
    int64_t at[WR];
    bool live[WR];
#pragma unroll
    for (int r = 0; r < WR; ++r) { // (clamped address, no branch around the load: the WR loads are in flight together)
        const int64_t seg = (row0 + r) * SPW + (lane >> LOGP);
        live[r] = seg < a.nseg && il < a.n;
        at[r] = live[r] ? seg * a.n + il : 0;
        x[r][0].k = load_raw<U, W>(a.in, at[r]);
    }
#pragma unroll
    for (int r = 0; r < WR; ++r) {
        x[r][0].k = live[r] ? to_ordered<U, W, KIND>(x[r][0].k, flip) : ~(U)0; // padding: behind every real key, every real position
        x[r][0].p = live[r] ? (uint32_t)il : 0x80000000u | (uint32_t)il;
    }
    net_from<U, P, 1, WR, 2, 1>(x);
#pragma unroll
    for (int r = 0; r < WR; ++r) {
        if (live[r]) { // (a sorted row keeps its real keys in slots 0 .. n - 1 of every segment)
            store_raw<U, W>(a.out, at[r], from_ordered<U, W, KIND>(x[r][0].k, flip));
            a.pos[at[r]] = (int64_t)x[r][0].p;
        }
    }
}

// Segments of 65 .. 512 keys: the same network with E = P / 64 slots per lane, one segment per wave: partner distances of 64 and
// more pair two registers of one lane (no exchange at all), the shorter ones go through the lanes as above; a direction bit at or
// above 64 is a constant of the register index.
template <typename U, int W, int KIND, int LOGP>
__global__ __launch_bounds__(256) void sort_wave_big_kernel(const SmallArgs a) {
    constexpr int P = 1 << LOGP, E = P / 64, WR = E == 2 ? 2 : 1;
    const int lane = threadIdx.x & 63;
    const int64_t seg0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * WR;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    WaveSlot<U> x[WR][E];
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int e = 0; e < E; ++e) { // (clamped address, no branch around the load)
            const int il = e * 64 + lane;
            const bool live = seg0 + r < a.nseg && il < a.n;
            x[r][e].k = load_raw<U, W>(a.in, live ? (seg0 + r) * a.n + il : 0);
        }
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int il = e * 64 + lane;
            const bool live = seg0 + r < a.nseg && il < a.n;
            x[r][e].k = live ? to_ordered<U, W, KIND>(x[r][e].k, flip) : ~(U)0; // padding: behind every real key, every real position
            x[r][e].p = live ? (uint32_t)il : 0x80000000u | (uint32_t)il;
        }
    net_from<U, P, E, WR, 2, 1>(x);
#pragma unroll
    for (int r = 0; r < WR; ++r)
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const int il = e * 64 + lane;
            if (seg0 + r < a.nseg && il < a.n) {
                const int64_t at = (seg0 + r) * a.n + il;
                store_raw<U, W>(a.out, at, from_ordered<U, W, KIND>(x[r][e].k, flip));
                a.pos[at] = (int64_t)x[r][e].p;
            }
        }
}

// The set of lanes (among those of m) that hold the same 8-bit digit as this lane: eight ballots, each folded into the running set by one
// three-input bit operation per half (m & ~(ballot ^ spread), spread = the lane's digit bit on all 32 positions): 4 VALU per bit.
__device__ __forceinline__ uint64_t match_digit8(uint32_t d, uint64_t m) {
    uint32_t lo = (uint32_t)m, hi = (uint32_t)(m >> 32);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
        uint32_t spread = (uint32_t)((int32_t)(d << (31 - b)) >> 31);
        asm volatile("" : "+v"(spread)); // (kept opaque: the ballot is taken from the spread bit, not from a second extraction of the digit)
        const uint64_t bb = __ballot(spread != 0);
        lo = __builtin_amdgcn_bitop3_b32(lo, (uint32_t)bb, spread, 0x90);
        hi = __builtin_amdgcn_bitop3_b32(hi, (uint32_t)(bb >> 32), spread, 0x90);
    }
    return ((uint64_t)hi << 32) | lo;
}

// ------------------------------------------------------------------------------------------
// mid-size segments: block-local LSD radix sort in LDS (one segment per block)
// ------------------------------------------------------------------------------------------
template <typename U, int W, int KIND, int NW, int ITEMS>
__global__ __launch_bounds__(NW * 64) void sort_block_radix_kernel(const SmallArgs a) {
    constexpr int NT = NW * 64, TILE = NT * ITEMS, WK = 64 * ITEMS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    U *skey = (U *)smem;                                            // [TILE]
    uint16_t *spos = (uint16_t *)(smem + (size_t)TILE * sizeof(U)); // [TILE]
    uint32_t(*cnt)[256] = (uint32_t(*)[256])(smem + (size_t)TILE * (sizeof(U) + 2));
    uint32_t *dstart = (uint32_t *)(cnt + NW), *wsum = dstart + 256;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int64_t segoff = (int64_t)blockIdx.x * a.n;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    const uint64_t below = (1ull << lane) - 1;
    U key[ITEMS];
    uint32_t pos[ITEMS], rank[ITEMS];
    // (no branch around a load: a slot behind the segment's end reads the segment's last key and is then replaced - all of a lane's loads are in
    //  flight before the first is waited for; with `i < n ? load : pad` the sixteen loads were sixteen round trips)
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) { // index order: wave w holds items w WK + r 64 + lane (the order the ranks below are taken in)
        const int i = w * WK + r * 64 + lane;
        key[r] = load_raw<U, W>(a.in, segoff + (i < a.n ? i : a.n - 1));
        pos[r] = (uint32_t)i;
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r)
        key[r] = (int)pos[r] < a.n ? to_ordered<U, W, KIND>(key[r], flip) : KeyBits<U, W>::all; // (padding: behind every real key)
#pragma unroll 1
    for (int pass = 0; pass < a.npass; ++pass) {
        const int shift = 8 * pass;
        for (int x = tid; x < NW * 256; x += NT) (&cnt[0][0])[x] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) { // rank among the wave's keys of the same digit: a popcount of the matching lanes below (stable)
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint64_t m = match_digit8(d, ~0ull);
            const uint32_t prev = cnt[w][d];
            rank[r] = prev + (uint32_t)__popcll(m & below);
            if ((m & below) == 0) cnt[w][d] = prev + (uint32_t)__popcll(m); // the lowest lane of the match set
        }
        __syncthreads();
        if (tid < 256) { // digit tid: the waves' counts -> exclusive offsets over the waves; exclusive scan of the digit totals
            uint32_t total = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) {
                const uint32_t c = cnt[i][tid];
                cnt[i][tid] = total;
                total += c;
            }
            uint32_t inc = total;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = __shfl_up(inc, o, 64);
                if (lane >= o) inc += up;
            }
            if (lane == 63) wsum[w] = inc;
            dstart[tid] = inc - total; // (+ the waves in front: below)
        }
        __syncthreads();
        if (tid < 256) {
            uint32_t off = dstart[tid];
            for (int i = 0; i < w; ++i) off += wsum[i];
#pragma unroll
            for (int i = 0; i < NW; ++i) cnt[i][tid] += off; // where wave i's keys of this digit start in the reordered segment
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t lp = cnt[w][d] + rank[r];
            skey[lp] = key[r];
            spos[lp] = (uint16_t)pos[r];
        }
        __syncthreads();
        if (pass + 1 < a.npass) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int i = w * WK + r * 64 + lane;
                key[r] = skey[i];
                pos[r] = spos[i];
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) { // (LDS reads of all of a lane's slots first, then the stores)
        const int i = tid + r * NT;
        key[r] = skey[i];
        pos[r] = spos[i];
    }
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) {
        const int i = tid + r * NT;
        if (i < a.n) {
            store_raw<U, W>(a.out, segoff + i, from_ordered<U, W, KIND>(key[r], flip));
            a.pos[segoff + i] = (int64_t)pos[r];
        }
    }
}
template <int NW, int ITEMS> static size_t block_radix_lds(size_t usz) { return (size_t)NW * 64 * ITEMS * (usz + 2) + (size_t)NW * 1024 + 1024 + 64; }

// ------------------------------------------------------------------------------------------
// long segments: LSD radix sort, 8-bit digits
// ------------------------------------------------------------------------------------------
// Tile shape by key width: threads per block, keys per thread (a wave owns 64 ITEMS consecutive keys). Measured, round 5, [1, 64 Mi] f32, one pass
// of the scatter kernel: 256 x 16 -> 254 us, 512 x 8 -> 239, 512 x 16 -> 231 (and half the tile counts to write, scan and read back).
#ifndef KF_SORT_R_NT
#define KF_SORT_R_NT 512
#endif
#ifndef KF_SORT_R_ITEMS
#define KF_SORT_R_ITEMS 16
#endif
#ifndef KF_SORT_R64_NT
#define KF_SORT_R64_NT 256
#endif
#ifndef KF_SORT_R64_ITEMS
#define KF_SORT_R64_ITEMS 16
#endif
template <typename U> struct RadixCfg;
template <> struct RadixCfg<uint32_t> { static constexpr int NT = KF_SORT_R_NT, ITEMS = KF_SORT_R_ITEMS, TILE = NT * ITEMS; };
template <> struct RadixCfg<uint64_t> { static constexpr int NT = KF_SORT_R64_NT, ITEMS = KF_SORT_R64_ITEMS, TILE = NT * ITEMS; };
#define KF_RADIX_CFG(U_) \
    constexpr int R_NT = RadixCfg<U_>::NT, R_ITEMS = RadixCfg<U_>::ITEMS, R_TILE = RadixCfg<U_>::TILE, R_WAVE_KEYS = 64 * R_ITEMS; \
    (void)R_NT; (void)R_ITEMS; (void)R_TILE; (void)R_WAVE_KEYS

struct RadixArgs {
    const void *src_keys; // typed keys (first pass) or ordered keys of U
    const uint32_t *src_pos;
    void *dst_keys; // ordered keys of U, or typed keys (last pass)
    void *dst_pos;  // uint32 positions, or int64 (last pass)
    uint32_t *counts;     // [nseg][ntiles][256]: tile histograms, then exclusive prefix over the tiles of a segment
    uint32_t *chunk_base; // [nseg][nchunks][256]: chunk totals, then exclusive prefix over the chunks of a segment
    uint32_t *digit_base; // [nseg][256]: exclusive prefix over the digits of a segment
    int64_t n;
    int ntiles, nchunks, shift, first, last, desc;
};

// Block -> tile. Blocks are dealt to the eight XCDs round-robin; every XCD takes one contiguous eighth of the tiles, so the runs that
// neighbouring tiles write behind each other meet in ONE L2 (round 5, [1, 64 Mi] f32: scatter 417 -> 359 us per pass; groups of 4 consecutive
// tiles per XCD do as well: 357).
__device__ __forceinline__ int radix_tile_of(int b, int ntiles) {
    const int q = ntiles >> 3, r = ntiles & 7, x = b & 7;
    return x * q + (x < r ? x : r) + (b >> 3);
}

template <typename U, int W, int KIND>
__device__ __forceinline__ U radix_load(const RadixArgs &a, int64_t at, U flip) {
    return a.first ? to_ordered<U, W, KIND>(load_raw<U, W>(a.src_keys, at), flip) : ((const U *)a.src_keys)[at];
}

// Tile histogram of one digit. The counters are kept in H_REP replicas picked by the lane (a replica's 256 counters start one bank further
// than the last one's): a digit that most keys share - the exponent byte of floats, the upper bytes of small integers - is then an
// 8-lane, not a 64-lane, serialisation of the LDS add. Whole tiles of 4- or 8-byte keys on a 16-byte boundary are read with 16-byte loads
// (the order inside a tile does not matter here).
constexpr int H_REP = 8, H_STRIDE = 257;

template <typename U, int W, int KIND>
__global__ __launch_bounds__(RadixCfg<U>::NT) void radix_hist_kernel(const RadixArgs a) {
    KF_RADIX_CFG(U);
    __shared__ uint32_t h[H_REP * H_STRIDE];
    const int tid = threadIdx.x;
    const int tile = radix_tile_of(blockIdx.x % a.ntiles, a.ntiles);
    const int64_t seg = blockIdx.x / a.ntiles, segoff = seg * a.n;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    for (int i = tid; i < H_REP * H_STRIDE; i += R_NT) h[i] = 0;
    __syncthreads();
    uint32_t *hr = h + (tid & (H_REP - 1)) * H_STRIDE;
    const int64_t base = (int64_t)tile * R_TILE;
    constexpr int ES = (int)sizeof(U);          // bytes of a stored key where the 16-byte path applies (typed keys of the same width, or ordered keys)
    constexpr int PER = 16 / ES, NV = R_ITEMS / PER;
    const bool same_width = !a.first || W == ES;
    if (same_width && base + R_TILE <= a.n && ((((uintptr_t)a.src_keys) + (size_t)(segoff + base) * ES) & 15) == 0) {
        const uint4 *src = (const uint4 *)((const char *)a.src_keys + (size_t)(segoff + base) * ES);
        uint4 v[NV];
#pragma unroll
        for (int r = 0; r < NV; ++r) v[r] = src[r * R_NT + tid];
#pragma unroll
        for (int r = 0; r < NV; ++r) {
            const uint32_t e[4] = {v[r].x, v[r].y, v[r].z, v[r].w};
#pragma unroll
            for (int c = 0; c < PER; ++c) {
                U raw;
                if constexpr (ES == 4) raw = (U)e[c];
                else raw = (U)(((uint64_t)e[2 * c + 1] << 32) | e[2 * c]);
                const U o = a.first ? to_ordered<U, W, KIND>(raw, flip) : raw;
                atomicAdd(&hr[(uint32_t)(o >> a.shift) & 255u], 1u);
            }
        }
    } else {
#pragma unroll 4
        for (int r = 0; r < R_ITEMS; ++r) {
            const int64_t i = base + r * R_NT + tid;
            if (i < a.n) atomicAdd(&hr[(uint32_t)(radix_load<U, W, KIND>(a, segoff + i, flip) >> a.shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t c = 0;
#pragma unroll
        for (int i = 0; i < H_REP; ++i) c += h[i * H_STRIDE + tid];
        a.counts[((size_t)seg * a.ntiles + tile) * 256 + tid] = c;
    }
}

// Exclusive prefix of the tile histograms, per digit, in two levels so that one very long segment is not one block's
// serial walk: (1) one block per chunk of R_CHUNK tiles turns its tiles' counts into prefixes inside the chunk and writes
// the chunk totals; (2) one block per segment (thread group g of 4 walks a quarter of the chunks for digit d = tid & 255)
// turns the chunk totals into prefixes over the chunks and scans the digit totals into digit_base.
constexpr int R_CHUNK = 64;

__global__ __launch_bounds__(256) void radix_scan_tiles_kernel(uint32_t *counts, uint32_t *chunk_sum, int ntiles, int nchunks) {
    const int d = threadIdx.x, chunk = blockIdx.x % nchunks;
    const int64_t seg = blockIdx.x / nchunks;
    const int t0 = chunk * R_CHUNK, nt = min(R_CHUNK, ntiles - t0);
    uint32_t *c = counts + ((size_t)seg * ntiles + t0) * 256 + d;
    uint32_t v[R_CHUNK]; // all of the chunk's counts of this digit are in flight at once: the walk is 64 loads deep, not 64 round trips long
#pragma unroll
    for (int t = 0; t < R_CHUNK; ++t) v[t] = t < nt ? c[(size_t)t * 256] : 0u;
    uint32_t run = 0;
#pragma unroll
    for (int t = 0; t < R_CHUNK; ++t) {
        const uint32_t x = v[t];
        v[t] = run;
        run += x;
    }
#pragma unroll
    for (int t = 0; t < R_CHUNK; ++t)
        if (t < nt) c[(size_t)t * 256] = v[t];
    chunk_sum[((size_t)seg * nchunks + chunk) * 256 + d] = run;
}

__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *counts, uint32_t *digit_base, int ntiles) {
    __shared__ uint32_t part[4][256];
    __shared__ uint32_t wsum[4];
    const int d = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int64_t seg = blockIdx.x;
    uint32_t *c = counts + (size_t)seg * ntiles * 256 + d;
    const int chunk = (ntiles + 3) / 4, t0 = min(g * chunk, ntiles), t1 = min(t0 + chunk, ntiles);
    constexpr int FAST = 64; // up to 256 chunks (64 Mi keys): each group's quarter sits in registers between the sum and the prefix
    uint32_t v[FAST];
    uint32_t sum = 0;
    if (chunk <= FAST) {
#pragma unroll
        for (int t = 0; t < FAST; ++t) v[t] = t0 + t < t1 ? c[(size_t)(t0 + t) * 256] : 0u;
#pragma unroll
        for (int t = 0; t < FAST; ++t) sum += v[t];
    } else {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) sum += c[(size_t)t * 256];
    }
    part[g][d] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int i = 0; i < g; ++i) run += part[i][d];
    if (chunk <= FAST) {
#pragma unroll
        for (int t = 0; t < FAST; ++t) {
            if (t0 + t < t1) c[(size_t)(t0 + t) * 256] = run;
            run += v[t];
        }
    } else {
#pragma unroll 8
        for (int t = t0; t < t1; ++t) {
            const uint32_t x = c[(size_t)t * 256];
            c[(size_t)t * 256] = run;
            run += x;
        }
    }
    // exclusive scan of the digit totals by group 0 (waves 0-3)
    uint32_t total = 0, inc = 0;
    if (g == 0) {
        total = part[0][d] + part[1][d] + part[2][d] + part[3][d];
        inc = total;
        const int lane = d & 63;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[d >> 6] = inc;
    }
    __syncthreads();
    if (g == 0) {
        uint32_t off = 0;
        for (int i = 0; i < (d >> 6); ++i) off += wsum[i];
        digit_base[(size_t)seg * 256 + d] = off + inc - total;
    }
}

// One tile of the stable scatter. FULL: every slot of the tile holds a key (all tiles of a segment but possibly its last one) - no validity
// ballots, compares or guarded LDS accesses on that path (round 5: 359 -> 277 us per pass together with the 4-VALU-per-bit match).
// (Tried and dropped, same run: keys and positions through ONE staging buffer one after the other - half the LDS, twice the tiles per CU,
//  two more barriers: 244 us against 239 / 231 for the one-round 512 x 8 / 512 x 16 shapes.)
template <typename U, int W, int KIND, bool FULL>
__device__ __forceinline__ void radix_scatter_tile(const RadixArgs &a, uint32_t (*cnt)[256], uint32_t *gbase, uint32_t *wsum, char *stage, int tile,
                                                   int64_t seg) {
    KF_RADIX_CFG(U);
    constexpr int NW = R_NT / 64;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int64_t segoff = seg * a.n, tile0 = segoff + (int64_t)tile * R_TILE; // (uniform)
    const int nv = FULL ? R_TILE : (int)(a.n - (int64_t)tile * R_TILE);
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    const uint32_t l0 = (uint32_t)(w * R_WAVE_KEYS + lane); // the tile-local index of this lane's first key; row r is l0 + 64 r
    const uint64_t below = (1ull << lane) - 1;

    U key[R_ITEMS];
    uint32_t pos[R_ITEMS], rank[R_ITEMS];
    // (the first / later-pass branch stays OUTSIDE the row loops: all of a lane's loads are in flight before the first one is waited for -
    //  with the branch inside the loop the first pass was sixteen round trips long: 348 us against 250 for the later passes)
    if (a.first) {
        const char *src = (const char *)a.src_keys + (size_t)tile0 * W;
#pragma unroll
        for (int r = 0; r < R_ITEMS; ++r) {
            const uint32_t l = l0 + r * 64;
            key[r] = (FULL || (int)l < nv) ? load_raw<U, W>(src, l) : (U)0;
            pos[r] = (uint32_t)tile * R_TILE + l;
        }
#pragma unroll
        for (int r = 0; r < R_ITEMS; ++r) key[r] = to_ordered<U, W, KIND>(key[r], flip);
    } else {
        const U *ksrc = (const U *)a.src_keys + tile0;
        const uint32_t *psrc = a.src_pos + tile0;
#pragma unroll
        for (int r = 0; r < R_ITEMS; ++r) {
            const uint32_t l = l0 + r * 64;
            const bool valid = FULL || (int)l < nv;
            key[r] = valid ? ksrc[l] : (U)0;
            pos[r] = valid ? psrc[l] : 0u;
        }
    }
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        const bool valid = FULL || (int)(l0 + r * 64) < nv;
        const uint32_t d = (uint32_t)(key[r] >> a.shift) & 255u;
        const uint64_t m = match_digit8(d, FULL ? ~0ull : __ballot(valid));
        const uint32_t prev = cnt[w][d];
        rank[r] = prev + (uint32_t)__popcll(m & below);
        if (valid && (m & below) == 0) cnt[w][d] = prev + (uint32_t)__popcll(m); // the lowest lane of the match set
    }
    __syncthreads();
    uint32_t total = 0, inc = 0;
    if (tid < 256) { // digit tid: the waves' counts -> exclusive offsets over the waves; exclusive scan of the digit totals
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const uint32_t c = cnt[i][tid];
            cnt[i][tid] = total;
            total += c;
        }
        inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[w] = inc;
    }
    __syncthreads();
    if (tid < 256) {
        uint32_t off = 0;
        for (int i = 0; i < w; ++i) off += wsum[i];
        const uint32_t ex = off + inc - total;
#pragma unroll
        for (int i = 0; i < NW; ++i) cnt[i][tid] += ex; // where wave i's keys of this digit start in the reordered tile
        gbase[tid] = a.counts[((size_t)seg * a.ntiles + tile) * 256 + tid] + (a.nchunks ? a.chunk_base[((size_t)seg * a.nchunks + tile / R_CHUNK) * 256 + tid] : 0u) +
                     a.digit_base[(size_t)seg * 256 + tid] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) { // (from here on rank[] is the key's slot in the reordered tile)
        const uint32_t d = (uint32_t)(key[r] >> a.shift) & 255u;
        rank[r] += cnt[w][d];
    }
    U *skey = (U *)stage;
    uint32_t *spos = (uint32_t *)(stage + (size_t)R_TILE * sizeof(U));
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        if (FULL || (int)(l0 + r * 64) < nv) {
            skey[rank[r]] = key[r];
            spos[rank[r]] = pos[r];
        }
    }
    __syncthreads();
    // (reads first, then the digit bases, then the stores - each step over all of the lane's slots, and the last-pass branch outside them)
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        const int i = tid + r * R_NT;
        key[r] = (FULL || i < nv) ? skey[i] : (U)0;
        pos[r] = (FULL || i < nv) ? spos[i] : 0u;
    }
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) rank[r] = gbase[(uint32_t)(key[r] >> a.shift) & 255u] + (uint32_t)(tid + r * R_NT);
    if (a.last) {
#pragma unroll
        for (int r = 0; r < R_ITEMS; ++r) {
            if (FULL || tid + r * R_NT < nv) {
                store_raw<U, W>(a.dst_keys, segoff + rank[r], from_ordered<U, W, KIND>(key[r], flip));
                ((int64_t *)a.dst_pos)[segoff + rank[r]] = (int64_t)pos[r];
            }
        }
    } else {
        U *kdst = (U *)a.dst_keys + segoff;
        uint32_t *pdst = (uint32_t *)a.dst_pos + segoff;
#pragma unroll
        for (int r = 0; r < R_ITEMS; ++r) {
            if (FULL || tid + r * R_NT < nv) {
                kdst[rank[r]] = key[r];
                pdst[rank[r]] = pos[r];
            }
        }
    }
}

template <typename U, int W, int KIND>
__global__ __launch_bounds__(RadixCfg<U>::NT) void radix_scatter_kernel(const RadixArgs a) {
    KF_RADIX_CFG(U);
    constexpr int NW = R_NT / 64;
    __shared__ uint32_t cnt[NW][256];
    __shared__ uint32_t gbase[256], wsum[4];
    extern __shared__ __attribute__((aligned(16))) char rsmem[]; // the tile in its new order (dynamic)
    const int tid = threadIdx.x;
    const int tile = radix_tile_of(blockIdx.x % a.ntiles, a.ntiles);
    const int64_t seg = blockIdx.x / a.ntiles;
#pragma unroll
    for (int i = 0; i < NW * 256 / R_NT; ++i) (&cnt[0][0])[i * R_NT + tid] = 0;
    __syncthreads();
    if ((int64_t)(tile + 1) * R_TILE <= a.n) radix_scatter_tile<U, W, KIND, true>(a, cnt, gbase, wsum, rsmem, tile, seg);
    else radix_scatter_tile<U, W, KIND, false>(a, cnt, gbase, wsum, rsmem, tile, seg);
}

constexpr int64_t kSmallMax = 8192;  // one block per segment (or several segments per block) up to here
constexpr int64_t kBitonicMax = 512; // ... by the bitonic network up to here, by radix passes in LDS above

struct SortPlan {
    bool small;
    int ntiles, nchunks;
    size_t key_bytes, pos_bytes, counts_bytes, chunk_bytes, base_bytes, total;
};
static inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
static SortPlan make_plan(int dtype, int64_t nseg, int64_t n) {
    SortPlan p{};
    p.small = n <= kSmallMax;
    if (p.small || nseg == 0) return p;
    const size_t usz = dtype_size(dtype) == 8 ? 8 : 4;
    const int64_t tile = usz == 8 ? RadixCfg<uint64_t>::TILE : RadixCfg<uint32_t>::TILE;
    p.ntiles = (int)((n + tile - 1) / tile);
    p.key_bytes = up256((size_t)nseg * n * usz);
    p.pos_bytes = up256((size_t)nseg * n * 4);
    p.nchunks = (p.ntiles + R_CHUNK - 1) / R_CHUNK;
    p.counts_bytes = up256((size_t)nseg * p.ntiles * 256 * 4);
    p.chunk_bytes = up256((size_t)nseg * p.nchunks * 256 * 4);
    p.base_bytes = up256((size_t)nseg * 256 * 4);
    p.total = 2 * p.key_bytes + 2 * p.pos_bytes + p.counts_bytes + p.chunk_bytes + p.base_bytes;
    return p;
}

template <typename U, int W, int KIND>
static int run_sort(const void *in, void *out, int64_t *pos, int64_t nseg, int64_t n, int desc, const SortPlan &p, char *ws, hipStream_t st, int npass) {
    if (npass <= 0 || npass > W) npass = W;
    if (p.small && n > kBitonicMax) { // one segment per block, radix passes in LDS
        SmallArgs a{in, out, pos, nseg, (int)n, 0, npass, desc};
        KF_REQUIRE(nseg <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many segments");
        KF_PROF("sort_radix_lds", st);
#define KF_BLOCK_RADIX(NW_, IT_)                                                                                                         \
    {                                                                                                                                    \
        const size_t lds = block_radix_lds<NW_, IT_>(sizeof(U));                                                                         \
        KF_ENSURE_LDS((sort_block_radix_kernel<U, W, KIND, NW_, IT_>), lds);                                                                 \
        sort_block_radix_kernel<U, W, KIND, NW_, IT_><<<(unsigned)nseg, NW_ * 64, lds, st>>>(a);                                          \
    }
        if (n <= 1024) KF_BLOCK_RADIX(4, 4)
        else if (n <= 4096) KF_BLOCK_RADIX(4, 16)
        else KF_BLOCK_RADIX(8, 16)
#undef KF_BLOCK_RADIX
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    if (p.small && n <= 64) { // rows of 64 slots in registers
        SmallArgs a{in, out, pos, nseg, (int)n, 0, npass, desc};
        while ((1 << a.logp) < n) ++a.logp;
        if (a.logp < 1) a.logp = 1; // (a one-key segment takes a two-slot cell)
        const int64_t rows = (nseg + (64 >> a.logp) - 1) / (64 >> a.logp), grid = (rows + 4 * kWaveRows - 1) / (4 * kWaveRows);
        KF_REQUIRE(grid <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many segments");
        KF_PROF("sort_bitonic_wave", st);
        switch (a.logp) {
        case 1: sort_wave_kernel<U, W, KIND, 1><<<(unsigned)grid, 256, 0, st>>>(a); break;
        case 2: sort_wave_kernel<U, W, KIND, 2><<<(unsigned)grid, 256, 0, st>>>(a); break;
        case 3: sort_wave_kernel<U, W, KIND, 3><<<(unsigned)grid, 256, 0, st>>>(a); break;
        case 4: sort_wave_kernel<U, W, KIND, 4><<<(unsigned)grid, 256, 0, st>>>(a); break;
        case 5: sort_wave_kernel<U, W, KIND, 5><<<(unsigned)grid, 256, 0, st>>>(a); break;
        default: sort_wave_kernel<U, W, KIND, 6><<<(unsigned)grid, 256, 0, st>>>(a); break;
        }
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    if (p.small && n <= kBitonicMax) { // one segment per wave, up to eight slots per lane
        SmallArgs a{in, out, pos, nseg, (int)n, 0, npass, desc};
        while ((1 << a.logp) < n) ++a.logp;
        const int wr = a.logp == 7 ? 2 : 1;
        const int64_t grid = (nseg + 4 * wr - 1) / (4 * wr);
        KF_REQUIRE(grid <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many segments");
        KF_PROF("sort_bitonic_wave", st);
        switch (a.logp) {
        case 7: sort_wave_big_kernel<U, W, KIND, 7><<<(unsigned)grid, 256, 0, st>>>(a); break;
        case 8: sort_wave_big_kernel<U, W, KIND, 8><<<(unsigned)grid, 256, 0, st>>>(a); break;
        default: sort_wave_big_kernel<U, W, KIND, 9><<<(unsigned)grid, 256, 0, st>>>(a); break;
        }
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    KF_RADIX_CFG(U);
    char *keys[2] = {ws, ws + p.key_bytes};
    char *poss[2] = {ws + 2 * p.key_bytes, ws + 2 * p.key_bytes + p.pos_bytes};
    uint32_t *counts = (uint32_t *)(ws + 2 * p.key_bytes + 2 * p.pos_bytes);
    uint32_t *cbase = (uint32_t *)((char *)counts + p.counts_bytes);
    uint32_t *dbase = (uint32_t *)((char *)cbase + p.chunk_bytes);
    const int64_t grid = nseg * p.ntiles;
    KF_REQUIRE(grid <= 0x7fffffff && nseg <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many tiles");
    KF_PROF("sort_radix", st);
    const bool one_level = p.ntiles <= 256;
    for (int pass = 0; pass < npass; ++pass) {
        RadixArgs a{};
        a.first = pass == 0;
        a.last = pass == npass - 1;
        a.src_keys = a.first ? in : keys[(pass - 1) & 1];
        a.src_pos = a.first ? nullptr : (const uint32_t *)poss[(pass - 1) & 1];
        a.dst_keys = a.last ? out : (void *)keys[pass & 1];
        a.dst_pos = a.last ? (void *)pos : (void *)poss[pass & 1];
        a.counts = counts;
        a.chunk_base = cbase;
        a.digit_base = dbase;
        a.n = n;
        a.ntiles = p.ntiles;
        a.nchunks = p.nchunks;
        a.shift = 8 * pass;
        a.desc = desc;
        radix_hist_kernel<U, W, KIND><<<(unsigned)grid, R_NT, 0, st>>>(a);
        if (one_level) { // up to 256 tiles per segment: the second-level kernel walks the tile counts themselves (one launch less per pass)
            a.nchunks = 0;
            radix_scan_kernel<<<(unsigned)nseg, 1024, 0, st>>>(counts, dbase, p.ntiles);
        } else {
            radix_scan_tiles_kernel<<<(unsigned)(nseg * p.nchunks), 256, 0, st>>>(counts, cbase, p.ntiles, p.nchunks);
            radix_scan_kernel<<<(unsigned)nseg, 1024, 0, st>>>(cbase, dbase, p.nchunks);
        }
        const size_t lds = (size_t)R_TILE * (sizeof(U) + 4);
        KF_ENSURE_LDS((radix_scatter_kernel<U, W, KIND>), lds);
        radix_scatter_kernel<U, W, KIND><<<(unsigned)grid, R_NT, lds, st>>>(a);
        KF_LAUNCH_CHECK();
    }
    return KF_OK;
}

} // namespace kf

using namespace kf;

extern "C" size_t kf_sort_workspace_bytes(int dtype, int64_t nseg, int64_t n) {
    if (nseg <= 0 || n <= 0 || dtype_size(dtype) == 0) return 0;
    return make_plan(dtype, nseg, n).total;
}

extern "C" int kf_sort(int dtype, const void *keys_in, void *keys_out, int64_t *pos_out, int64_t nseg, int64_t n, int descending,
                       void *workspace, size_t workspace_bytes, void *stream) {
    return kf::sort_with_key_bits(dtype, keys_in, keys_out, pos_out, nseg, n, descending, workspace, workspace_bytes, stream, 0);
}

// key_bits > 0: the caller knows that all keys agree above their low key_bits bits (row numbers below a table's row count): the radix
// passes over the bytes above them are skipped - the same result, fewer launches (the network for segments of up to 512 keys ignores it).
int kf::sort_with_key_bits(int dtype, const void *keys_in, void *keys_out, int64_t *pos_out, int64_t nseg, int64_t n, int descending, void *workspace,
                           size_t workspace_bytes, void *stream, int key_bits) {
    const int npass = key_bits > 0 ? (key_bits + 7) / 8 : 0;
    KF_REQUIRE(dtype != KF_BOOL && dtype_size(dtype) != 0, KF_ERR_UNSUPPORTED, "kf_sort: dtype %d not supported (bool cannot be sorted)", dtype);
    KF_REQUIRE(nseg >= 0 && n >= 0, KF_ERR_INVALID, "kf_sort: negative extent");
    KF_REQUIRE(n <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: a segment can not have more than INT_MAX elements");
    if (nseg == 0 || n == 0) return KF_OK;
    KF_REQUIRE(keys_in && keys_out && pos_out, KF_ERR_INVALID, "kf_sort: null buffer");
    KF_REQUIRE(keys_in != keys_out, KF_ERR_INVALID, "kf_sort: keys_in and keys_out must be different buffers");
    const SortPlan p = make_plan(dtype, nseg, n);
    KF_REQUIRE(p.total == 0 || (workspace && workspace_bytes >= p.total && (uintptr_t)workspace % 16 == 0), KF_ERR_WORKSPACE,
               "kf_sort: workspace of %zu bytes (16-B aligned) required, got %zu", p.total, workspace_bytes);
    hipStream_t st = as_stream(stream);
    char *ws = (char *)workspace;
    const int desc = descending != 0;
    switch (dtype) {
    case KF_U8: return run_sort<uint32_t, 1, K_UNSIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_I8: return run_sort<uint32_t, 1, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_I16: return run_sort<uint32_t, 2, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_F16: case KF_BF16: return run_sort<uint32_t, 2, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_I32: return run_sort<uint32_t, 4, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_F32: return run_sort<uint32_t, 4, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_I64: return run_sort<uint64_t, 8, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    case KF_F64: return run_sort<uint64_t, 8, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st, npass);
    default: break;
    }
    KF_REQUIRE(false, KF_ERR_UNSUPPORTED, "kf_sort: dtype %d not supported", dtype);
}
