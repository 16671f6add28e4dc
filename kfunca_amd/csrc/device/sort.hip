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
//   * n <= 512: one block sorts spb segments in LDS with a bitonic network over (key, position) composites -
//     positions are unique, so the total order IS the stable order and no ranking pass is needed. Short segments share a
//     block (2048 slots per block), so [68185 segments x 13 keys] does not launch 68185 nearly empty blocks.
//   * 512 < n <= 8192 (round 5): ONE block sorts ONE segment by a least-significant-digit radix sort that never leaves LDS - 8-bit digits, one
//     pass per key byte, the ballot ranking of the long-segment path below, positions carried as 16-bit values. The bitonic network needs
//     log2(P) (log2(P) + 1) / 2 = 78 LDS passes over a 4096-key segment (40 k cycles of LDS pipe); four radix passes need ~10 k cycles of VALU.
//     Padding slots hold the largest key and the largest positions: a stable sort leaves them behind every real key.
//   * n > 8192: least-significant-digit radix sort, 8-bit digits (one pass per key byte), four launches per pass:
//     tile histograms -> two-level exclusive scan (inside chunks of 64 tiles, then over the chunks) -> stable scatter. A tile is 4096 keys, 1024 consecutive keys per wave.
//     Ranking inside a wave is by digit match: 8 ballots give each lane the set of lanes holding its digit, the lane's
//     rank is a popcount below itself - no per-key atomics, stable by construction. The tile is first reordered in LDS,
//     then written out, so every digit's run leaves as consecutive addresses. The first pass reads the caller's typed
//     keys, the last writes typed keys and int64 positions; the passes in between ping-pong through the workspace.
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

// ------------------------------------------------------------------------------------------
// short segments: bitonic network in LDS
// ------------------------------------------------------------------------------------------
template <typename U> struct Slots;
template <> struct Slots<uint32_t> { // key and position in one 64-bit word: one compare
    uint64_t *c;
    struct V { uint64_t v; };
    __device__ void init(char *smem, int) { c = (uint64_t *)smem; }
    __device__ V get(int i) const { return {c[i]}; }
    __device__ void put(int i, V x) { c[i] = x.v; }
    static __device__ V make(uint32_t k, uint32_t i) { return {((uint64_t)k << 32) | i}; }
    static __device__ bool lt(V a, V b) { return a.v < b.v; }
    static __device__ uint32_t key(V a) { return (uint32_t)(a.v >> 32); }
    static __device__ uint32_t pos(V a) { return (uint32_t)a.v; }
    static size_t bytes(int e) { return (size_t)e * 8; }
};
template <> struct Slots<uint64_t> {
    uint64_t *k;
    uint32_t *p;
    struct V { uint64_t k; uint32_t p; };
    __device__ void init(char *smem, int e) { k = (uint64_t *)smem; p = (uint32_t *)(smem + (size_t)e * 8); }
    __device__ V get(int i) const { return {k[i], p[i]}; }
    __device__ void put(int i, V x) { k[i] = x.k; p[i] = x.p; }
    static __device__ V make(uint64_t k, uint32_t i) { return {k, i}; }
    static __device__ bool lt(V a, V b) { return a.k < b.k || (a.k == b.k && a.p < b.p); }
    static __device__ uint64_t key(V a) { return a.k; }
    static __device__ uint32_t pos(V a) { return a.p; }
    static size_t bytes(int e) { return (size_t)e * 12; }
};

struct SmallArgs {
    const void *in;
    void *out;
    int64_t *pos;
    int64_t nseg;
    int n, logp, spb, desc;
};

template <typename U, int W, int KIND>
__global__ __launch_bounds__(1024) void sort_small_kernel(const SmallArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int P = 1 << a.logp, E = a.spb << a.logp, nthr = blockDim.x, tid = threadIdx.x;
    Slots<U> s;
    s.init(smem, E);
    using V = typename Slots<U>::V;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
    const int64_t seg0 = (int64_t)blockIdx.x * a.spb;
    for (int e = tid; e < E; e += nthr) {
        const int il = e & (P - 1);
        const int64_t seg = seg0 + (e >> a.logp);
        V v = Slots<U>::make(~(U)0, 0x80000000u | (uint32_t)il); // padding: behind every real key, every real position
        if (seg < a.nseg && il < a.n) v = Slots<U>::make(to_ordered<U, W, KIND>(load_raw<U, W>(a.in, seg * a.n + il), flip), (uint32_t)il);
        s.put(e, v);
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (E >> 1); t += nthr) {
                const int i = 2 * t - (t & (j - 1)), l = i + j;
                const bool up = ((i & (P - 1)) & k) == 0;
                const V x = s.get(i), y = s.get(l);
                if (Slots<U>::lt(y, x) == up) {
                    s.put(i, y);
                    s.put(l, x);
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < E; e += nthr) {
        const int il = e & (P - 1);
        const int64_t seg = seg0 + (e >> a.logp);
        if (seg < a.nseg && il < a.n) {
            const V v = s.get(e);
            store_raw<U, W>(a.out, seg * a.n + il, from_ordered<U, W, KIND>(Slots<U>::key(v), flip));
            a.pos[seg * a.n + il] = (int64_t)Slots<U>::pos(v);
        }
    }
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
#pragma unroll
    for (int r = 0; r < ITEMS; ++r) { // index order: wave w holds items w WK + r 64 + lane (the order the ranks below are taken in)
        const int i = w * WK + r * 64 + lane;
        key[r] = i < a.n ? to_ordered<U, W, KIND>(load_raw<U, W>(a.in, segoff + i), flip) : KeyBits<U, W>::all; // (padding: behind every real key)
        pos[r] = (uint32_t)i;
    }
#pragma unroll 1
    for (int pass = 0; pass < W; ++pass) {
        const int shift = 8 * pass;
        for (int x = tid; x < NW * 256; x += NT) (&cnt[0][0])[x] = 0;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) { // rank among the wave's keys of the same digit: a popcount of the matching lanes below (stable)
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            uint64_t m = ~0ull;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const bool bit = (d >> b) & 1u;
                const uint64_t bb = __ballot(bit);
                m &= bit ? bb : ~bb;
            }
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
            uint32_t off = 0;
            for (int i = 0; i < w; ++i) off += wsum[i];
            dstart[tid] += off;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < ITEMS; ++r) {
            const uint32_t d = (uint32_t)(key[r] >> shift) & 255u;
            const uint32_t lp = dstart[d] + cnt[w][d] + rank[r];
            skey[lp] = key[r];
            spos[lp] = (uint16_t)pos[r];
        }
        __syncthreads();
        if (pass + 1 < W) {
#pragma unroll
            for (int r = 0; r < ITEMS; ++r) {
                const int i = w * WK + r * 64 + lane;
                key[r] = skey[i];
                pos[r] = spos[i];
            }
            __syncthreads();
        }
    }
    for (int i = tid; i < a.n; i += NT) {
        store_raw<U, W>(a.out, segoff + i, from_ordered<U, W, KIND>(skey[i], flip));
        a.pos[segoff + i] = (int64_t)spos[i];
    }
}
template <int NW, int ITEMS> static size_t block_radix_lds(size_t usz) { return (size_t)NW * 64 * ITEMS * (usz + 2) + (size_t)NW * 1024 + 1024 + 64; }

// ------------------------------------------------------------------------------------------
// long segments: LSD radix sort, 8-bit digits
// ------------------------------------------------------------------------------------------
#ifndef KF_SORT_R_ITEMS
#define KF_SORT_R_ITEMS 16 // keys per thread. Measured (round 5, [1, 64 Mi] f32): 16 -> 2.22 ms, 32 (8192-key tiles: longer runs, half the blocks per CU) -> 2.85 ms
#endif
constexpr int R_NT = 256, R_ITEMS = KF_SORT_R_ITEMS, R_TILE = R_NT * R_ITEMS, R_WAVE_KEYS = 64 * R_ITEMS;

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

template <typename U, int W, int KIND>
__device__ __forceinline__ U radix_load(const RadixArgs &a, int64_t at, U flip) {
    return a.first ? to_ordered<U, W, KIND>(load_raw<U, W>(a.src_keys, at), flip) : ((const U *)a.src_keys)[at];
}

template <typename U, int W, int KIND>
__global__ __launch_bounds__(R_NT) void radix_hist_kernel(const RadixArgs a) {
    __shared__ uint32_t h[R_NT / 64][256];
    const int tid = threadIdx.x, w = tid >> 6;
    const int tile = blockIdx.x % a.ntiles;
    const int64_t seg = blockIdx.x / a.ntiles, segoff = seg * a.n;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
#pragma unroll
    for (int i = 0; i < R_NT / 64; ++i) h[i][tid] = 0;
    __syncthreads();
    const int64_t base = (int64_t)tile * R_TILE;
#pragma unroll 4
    for (int r = 0; r < R_ITEMS; ++r) {
        const int64_t i = base + r * R_NT + tid;
        if (i < a.n) atomicAdd(&h[w][(uint32_t)(radix_load<U, W, KIND>(a, segoff + i, flip) >> a.shift) & 255u], 1u);
    }
    __syncthreads();
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < R_NT / 64; ++i) c += h[i][tid];
    a.counts[((size_t)seg * a.ntiles + tile) * 256 + tid] = c;
}

// Exclusive prefix of the tile histograms, per digit, in two levels so that one very long segment is not one block's
// serial walk: (1) one block per chunk of R_CHUNK tiles turns its tiles' counts into prefixes inside the chunk and writes
// the chunk totals; (2) one block per segment (thread group g of 4 walks a quarter of the chunks for digit d = tid & 255)
// turns the chunk totals into prefixes over the chunks and scans the digit totals into digit_base.
constexpr int R_CHUNK = 64;

__global__ __launch_bounds__(256) void radix_scan_tiles_kernel(uint32_t *counts, uint32_t *chunk_sum, int ntiles, int nchunks) {
    const int d = threadIdx.x, chunk = blockIdx.x % nchunks;
    const int64_t seg = blockIdx.x / nchunks;
    const int t0 = chunk * R_CHUNK, t1 = min(t0 + R_CHUNK, ntiles);
    uint32_t *c = counts + ((size_t)seg * ntiles + t0) * 256 + d;
    uint32_t run = 0;
#pragma unroll 8
    for (int t = 0; t < t1 - t0; ++t) {
        const uint32_t v = c[(size_t)t * 256];
        c[(size_t)t * 256] = run;
        run += v;
    }
    chunk_sum[((size_t)seg * nchunks + chunk) * 256 + d] = run;
}

__global__ __launch_bounds__(1024) void radix_scan_kernel(uint32_t *counts, uint32_t *digit_base, int ntiles) {
    __shared__ uint32_t part[4][256];
    __shared__ uint32_t wsum[4];
    const int d = threadIdx.x & 255, g = threadIdx.x >> 8;
    const int64_t seg = blockIdx.x;
    uint32_t *c = counts + (size_t)seg * ntiles * 256 + d;
    const int chunk = (ntiles + 3) / 4, t0 = min(g * chunk, ntiles), t1 = min(t0 + chunk, ntiles);
    uint32_t sum = 0;
#pragma unroll 8
    for (int t = t0; t < t1; ++t) sum += c[(size_t)t * 256];
    part[g][d] = sum;
    __syncthreads();
    uint32_t run = 0;
    for (int i = 0; i < g; ++i) run += part[i][d];
#pragma unroll 8
    for (int t = t0; t < t1; ++t) {
        const uint32_t v = c[(size_t)t * 256];
        c[(size_t)t * 256] = run;
        run += v;
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

template <typename U, int W, int KIND>
__global__ __launch_bounds__(R_NT) void radix_scatter_kernel(const RadixArgs a) {
    __shared__ uint32_t cnt[R_NT / 64][256];
    __shared__ uint32_t dstart[256], gbase[256], wsum[R_NT / 64];
    extern __shared__ __attribute__((aligned(16))) char rsmem[]; // the tile in its new order: keys, then positions (dynamic: 64 - 96 KiB)
    U *skey = (U *)rsmem;
    uint32_t *spos = (uint32_t *)(rsmem + (size_t)R_TILE * sizeof(U));
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int tile = blockIdx.x % a.ntiles;
    const int64_t seg = blockIdx.x / a.ntiles, segoff = seg * a.n;
    const U flip = a.desc ? KeyBits<U, W>::all : (U)0;
#pragma unroll
    for (int i = 0; i < R_NT / 64; ++i) cnt[i][tid] = 0;
    __syncthreads();

    U key[R_ITEMS];
    uint32_t pos[R_ITEMS], rank[R_ITEMS];
    const int64_t wbase = (int64_t)tile * R_TILE + w * R_WAVE_KEYS + lane;
    const uint64_t below = (1ull << lane) - 1;
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        const int64_t i = wbase + r * 64;
        const bool valid = i < a.n;
        key[r] = valid ? radix_load<U, W, KIND>(a, segoff + i, flip) : (U)0;
        pos[r] = valid ? (a.first ? (uint32_t)i : a.src_pos[segoff + i]) : 0u;
    }
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        const bool valid = wbase + r * 64 < a.n;
        const uint32_t d = (uint32_t)(key[r] >> a.shift) & 255u;
        uint64_t m = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bb = __ballot(bit);
            m &= bit ? bb : ~bb;
        }
        const uint32_t prev = cnt[w][d];
        rank[r] = prev + (uint32_t)__popcll(m & below);
        if (valid && (m & below) == 0) cnt[w][d] = prev + (uint32_t)__popcll(m); // the lowest lane of the match set
    }
    __syncthreads();
    {
        uint32_t c[R_NT / 64], total = 0;
#pragma unroll
        for (int i = 0; i < R_NT / 64; ++i) {
            c[i] = cnt[i][tid];
            cnt[i][tid] = total;
            total += c[i];
        }
        uint32_t inc = total;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o, 64);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        uint32_t off = 0;
        for (int i = 0; i < w; ++i) off += wsum[i];
        const uint32_t ex = off + inc - total;
        dstart[tid] = ex;
        gbase[tid] = a.counts[((size_t)seg * a.ntiles + tile) * 256 + tid] + a.chunk_base[((size_t)seg * a.nchunks + tile / R_CHUNK) * 256 + tid] +
                     a.digit_base[(size_t)seg * 256 + tid] - ex;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < R_ITEMS; ++r) {
        if (wbase + r * 64 < a.n) {
            const uint32_t d = (uint32_t)(key[r] >> a.shift) & 255u;
            const uint32_t lp = dstart[d] + cnt[w][d] + rank[r];
            skey[lp] = key[r];
            spos[lp] = pos[r];
        }
    }
    __syncthreads();
    const int64_t left = a.n - (int64_t)tile * R_TILE;
    const int nv = left < R_TILE ? (int)left : R_TILE;
    for (int i = tid; i < nv; i += R_NT) {
        const U k = skey[i];
        const int64_t at = segoff + (uint32_t)(gbase[(uint32_t)(k >> a.shift) & 255u] + (uint32_t)i);
        if (a.last) {
            store_raw<U, W>(a.dst_keys, at, from_ordered<U, W, KIND>(k, flip));
            ((int64_t *)a.dst_pos)[at] = (int64_t)spos[i];
        } else {
            ((U *)a.dst_keys)[at] = k;
            ((uint32_t *)a.dst_pos)[at] = spos[i];
        }
    }
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
    p.ntiles = (int)((n + R_TILE - 1) / R_TILE);
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
static int run_sort(const void *in, void *out, int64_t *pos, int64_t nseg, int64_t n, int desc, const SortPlan &p, char *ws, hipStream_t st) {
    if (p.small && n > kBitonicMax) { // one segment per block, radix passes in LDS
        SmallArgs a{in, out, pos, nseg, (int)n, 0, 1, desc};
        KF_REQUIRE(nseg <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many segments");
        KF_PROF("sort_radix_lds", st);
#define KF_BLOCK_RADIX(NW_, IT_)                                                                                                         \
    {                                                                                                                                    \
        const size_t lds = block_radix_lds<NW_, IT_>(sizeof(U));                                                                         \
        KF_HIP_TRY(hipFuncSetAttribute((const void *)sort_block_radix_kernel<U, W, KIND, NW_, IT_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        sort_block_radix_kernel<U, W, KIND, NW_, IT_><<<(unsigned)nseg, NW_ * 64, lds, st>>>(a);                                          \
    }
        if (n <= 1024) KF_BLOCK_RADIX(4, 4)
        else if (n <= 4096) KF_BLOCK_RADIX(4, 16)
        else KF_BLOCK_RADIX(8, 16)
#undef KF_BLOCK_RADIX
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    if (p.small) {
        SmallArgs a{in, out, pos, nseg, (int)n, 0, 1, desc};
        while ((1 << a.logp) < n) ++a.logp;
        if (a.logp < 1) a.logp = 1;
        const int P = 1 << a.logp;
        a.spb = P >= 2048 ? 1 : 2048 / P;
        if ((int64_t)a.spb > nseg) a.spb = (int)nseg;
        const int E = a.spb * P;
        int nthr = E / 2;
        nthr = nthr < 64 ? 64 : (nthr > 1024 ? 1024 : (nthr + 63) / 64 * 64);
        const size_t lds = Slots<U>::bytes(E);
        KF_HIP_TRY(hipFuncSetAttribute((const void *)sort_small_kernel<U, W, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)Slots<U>::bytes((int)kSmallMax)));
        const int64_t grid = (nseg + a.spb - 1) / a.spb;
        KF_REQUIRE(grid <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many segments");
        KF_PROF("sort_bitonic_lds", st);
        sort_small_kernel<U, W, KIND><<<(unsigned)grid, nthr, lds, st>>>(a);
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    char *keys[2] = {ws, ws + p.key_bytes};
    char *poss[2] = {ws + 2 * p.key_bytes, ws + 2 * p.key_bytes + p.pos_bytes};
    uint32_t *counts = (uint32_t *)(ws + 2 * p.key_bytes + 2 * p.pos_bytes);
    uint32_t *cbase = (uint32_t *)((char *)counts + p.counts_bytes);
    uint32_t *dbase = (uint32_t *)((char *)cbase + p.chunk_bytes);
    const int64_t grid = nseg * p.ntiles;
    KF_REQUIRE(grid <= 0x7fffffff && nseg <= 0x7fffffff, KF_ERR_INDEX_RANGE, "kf_sort: too many tiles");
    KF_PROF("sort_radix", st);
    for (int pass = 0; pass < W; ++pass) {
        RadixArgs a{};
        a.first = pass == 0;
        a.last = pass == W - 1;
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
        radix_scan_tiles_kernel<<<(unsigned)(nseg * p.nchunks), 256, 0, st>>>(counts, cbase, p.ntiles, p.nchunks);
        radix_scan_kernel<<<(unsigned)nseg, 1024, 0, st>>>(cbase, dbase, p.nchunks);
        const size_t lds = (size_t)R_TILE * (sizeof(U) + 4);
        KF_HIP_TRY(hipFuncSetAttribute((const void *)radix_scatter_kernel<U, W, KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
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
    case KF_U8: return run_sort<uint32_t, 1, K_UNSIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_I8: return run_sort<uint32_t, 1, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_I16: return run_sort<uint32_t, 2, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_F16: case KF_BF16: return run_sort<uint32_t, 2, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_I32: return run_sort<uint32_t, 4, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_F32: return run_sort<uint32_t, 4, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_I64: return run_sort<uint64_t, 8, K_SIGNED>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    case KF_F64: return run_sort<uint64_t, 8, K_FLOAT>(keys_in, keys_out, pos_out, nseg, n, desc, p, ws, st);
    default: break;
    }
    KF_REQUIRE(false, KF_ERR_UNSUPPORTED, "kf_sort: dtype %d not supported", dtype);
}
