// Causal attention forward + backward for gfx950.
//
// Replaces src/device/causal_attention_kernel.cu:9-72 (+ utils/causal_attention.h,
// causal_attention_ref.h, block_utils.h). Semantics are the reference's ref kernel
// (causal_attention_ref.h:25-64): scores = Q K^T / sqrt(D); key n is visible to query m iff
// m >= n (absolute indices, top-left aligned); max-subtracted softmax; O = P V. The reference
// has no backward and allocates an O(S^2) scratch on every call (causal_attention_kernel.cu:22);
// neither is reproduced: the forward also emits LSE = m + log(l) per query so the backward can
// recompute P, and nothing of size S^2 is ever materialised.
//
// Two implementations behind one ABI:
//  * MFMA path (bf16/f16, D = 64 or 128 - the reference's two fast head sizes, causal_attention_kernel.cu:25-60 -, Sq % 128 == 0,
//    Skv % 128 == 0; every kernel of it is a template on the head size): flash-style, everything kept in
//    the "query/key on the lane" orientation so softmax statistics are lane-local:
//      S^T = K Q^T      A = K rows (LDS, ds_read_b128), B = Q fragments (registers)
//      O^T += V^T P^T   A = V^T via ds_read_b64_tr_b16 (hardware transpose read of the row-major
//                        LDS tile), B = the S^T accumulator converted in place to bf16
//    K/V tiles use one XOR-swizzled LDS image that is conflict-free for both the row reads and
//    the transposed reads; causal tiles above the diagonal are skipped.
//    Backward = delta pre-pass + a dK/dV kernel (key on the lane: S, dP, dV^T += dO^T P, dK^T += Q^T dS) that also STORES
//    dS = P o (dP - delta) in 16 bits + a dQ kernel that streams it back (dQ^T += K^T dS^T): the 5 matrix products of the
//    algorithm, no atomics (dQ, dK, dV bitwise reproducible), one 16-bit S x S / 2 round trip through HBM. A recomputing dQ
//    kernel (S and dP again: 7 products, small workspace) remains behind KF_ATTN_SPLIT_BWD and for dS beyond 64 GiB.
//    Operands may be strided (batch / head / row strides, kf_attn_*_strided): q, k, v in place inside a packed QKV projection.
//    A workgroup takes a block and its causal mirror, so all workgroups carry the same work (a_block_map, `persist`).
//  * f32 forward (the reference's own dtype and fast path, D = 64 | 128): exact-f32 MFMA kernel.
//  * generic path (any other dtype / ragged shape / head size <= 256, and the f32 backward):
//    LDS-tiled f32 online-softmax kernel; backward by two LDS-tiled f32 kernels.
#include <math.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace kf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

struct AttnArgs {
    const char *q, *k, *v, *o, *d_o;
    char *out, *dq, *dk, *dv;
    float *lse;        // forward: written; backward: read
    const float *lse_r;
    float *delta;
    float *nlse, *ndelta; // backward, dK/dV v4: -lse * sqrt(D) and -delta (initial accumulators of S and dP)
    int64_t B, H, Sq, Skv, D;
    int64_t Sqc;       // backward, 16-bit matrix-core path: rows per head of the two row-constant arrays nlse / ndelta = Sq rounded up to 32 (a slice), pad rows zero
    float scale;
    float scale_log2e; // scale * log2(e), formed on the host (attn_fwd_w4_kernel hands it to its instruction stream as a scalar)
    int xcd_map; // 1: nbh % 8 == 0, heads are pinned to XCDs (a_block_map)
    float defer; // forward: adopt a new running maximum only beyond this many exponent units (kDeferMax; -inf: always)
    unsigned nvwg;   // the generated kernels: virtual workgroups (= gridDim.x unless KF_ATTN_GRID_WGS asks for fewer real ones: grid-stride loop)
    int persist;     // k > 0: a workgroup handles k pairs {block x, its causal mirror}: equal work per workgroup (k = 1 is used)
    int persist_rev; // the short block of a pair first
    // global layouts of the 16-bit matrix-core path: byte strides of batch, head and row (the last dim is contiguous). Contiguous
    // [B,H,S,D] tensors: {H S 256, S 256, 256}; q / k / v living inside one packed [B S, 3 H D] projection: {S 3 H D 2, 256, 3 H D 2}.
    struct Lay { int64_t sb, sh, sr; } lq, lk, lv, lo, ldo, ldq, ldk, ldv;
#ifdef KF_MUTANT
    int mutant;      // mutation build only (tests/test_gpu_attention_mutants.py): which deliberate defect is switched on
#endif
    int64_t bh0;     // backward, dS form: this launch covers the (batch, head) pairs bh0 .. bh0 + nbh - 1 (one group of the workspace cap)
    int nbh;         // pairs in this launch (forward and the other forms: B H, bh0 = 0)
    char *ds;        // backward: dS = P o (dP - delta) in 16 bits, written by the dK/dV kernel, read by the dQ kernel (null: not kept)
    int64_t ds_nkb, ds_pair; // its tile grid (ds_tile_index below): 256-key blocks; tiles per (batch, head) pair
    int ds_tri;              // 1: rows cut off behind their diagonal square (the causal half: half the bytes); 0: full rows (the rectangle of rounds 2-5)
#if defined(KF_FWD_W4_STAMPS) || defined(KF_DKV_W4_STAMPS)
    unsigned long long *dbg; // diagnostic builds (tools/attn_fwd_w4_timeline.py, attn_dkv_w4_timeline.py): where every wave writes its cycle sums
#endif
};

// dS workspace (backward): the dK/dV kernel already holds dS = P o (dP - delta) as packed 16-bit MFMA operands; it stores them
// and the dQ kernel computes dQ = scale dS K from them - 2 matrix products instead of the 6 a recomputing dQ kernel executes
// (S and dP again), at the price of one 16-bit S x S / 2 round trip through HBM (4.3 GB at B 8, H 32, S 4096: HBM-bound at ~0.8 ms).
// Layout: tiles of 32 keys x 32 queries (2 KiB), per (batch, head) pair ordered [256-query block qb][256-key block kb][32-key block of kb: 0..7][slice of qb:
// 0..7]. Two forms (ds_tri_mode below picks): FULL ROWS of nkb squares of 64 tiles (rounds 2-5: the rectangle, its upper half never touched), or - round 6,
// VERDICT round 5 next #8 - THE CAUSAL HALF: every row cut off behind its diagonal square, row qb holds min(qb + 1, nkb) squares.
// What a dQ workgroup (one query block) reads is still ONE contiguous stream, 32 KiB per 64-key step; a dK/dV wave's tile of the next
// slice is 2 KiB on, the next query block's a row further (a row grows by one square per query block: the stream keeps the step in a
// scalar register). Query blocks are whole (a dQ wave without queries still fetches its tiles).
// (Measured same-box before settling on this: [kb][slice][32-key block] - a constant pointer step for dK/dV - and [kb][qb][..] both cost
// the dQ kernel 4-9 %: its stream then jumps between sixteen areas of the pair instead of walking one row. The causal half against full rows under the
// same code: no difference in tools/attn_bench.py's loop (0.929 vs 0.939 ms), 3-12 % slower dQ inside bench.py's step (rows no longer start on 2 MiB
// boundaries; the pair stride is 17 MiB): profiles/r06_ab_ds_layout.txt - hence the two forms.)
// Inside a tile: [s][key][hl][8 values] where the 8 values are accumulator registers e = 8 s + j of lane half hl, i.e. queries
// (j & 3) + 8 (2 s + (j >> 2)) + 4 hl of the slice - exactly one packed operand of the dK/dV wave (key on the lane), so a store
// instruction writes 1 KiB of consecutive bytes.
constexpr int DS_TILE = 2048;
// tiles in front of query block qb's row within one pair. tri: sum over j < qb of min(j + 1, nkb) squares of 64 tiles; else qb rows of nkb squares
__host__ __device__ inline int64_t ds_row_base(int64_t qb, int64_t nkb, int tri) {
    if (!tri) return qb * nkb * 64;
    const int64_t m = qb < nkb ? qb : nkb;
    return 32 * m * (m + 1) + (qb - m) * nkb * 64;
}
__host__ __device__ inline int64_t ds_pair_tiles(int64_t Sq, int64_t Skv, int tri) { return ds_row_base((Sq + 255) / 256, (Skv + 255) / 256, tri); }
__host__ __device__ inline size_t ds_bytes(int64_t nbh, int64_t Sq, int64_t Skv, int tri) { return (size_t)nbh * (size_t)ds_pair_tiles(Sq, Skv, tri) * DS_TILE; }
// tile (kwb = 32-key block, sl = 32-query slice) of a pair, kwb / 8 <= sl / 8
__host__ __device__ inline int64_t ds_tile_index(int64_t kwb, int64_t sl, int64_t nkb, int tri) { return ds_row_base(sl >> 3, nkb, tri) + kwb * 8 + (sl & 7); }

// XCD-aware block order for the v2 kernels (1-D grid of nx * nbh blocks). Hardware deals block ids round-robin
// over the 8 XCDs, each with a private 4 MiB L2. All nx blocks of one (batch, head) re-read that head's K/V
// (or Q/dO): 2 MiB at S = 4096. Dealing a head's blocks to ONE XCD keeps those re-reads in its L2 instead of
// fetching every head into every L2. Speed only: any placement is correct.
// base of (batch, head) bh = b * H + h under a layout
__device__ __forceinline__ int64_t a_head(const AttnArgs::Lay &l, int64_t bh, int64_t H) { return (bh / H) * l.sb + (bh % H) * l.sh; }

__device__ __forceinline__ void a_block_map(int nx, int nbh, int xcd_map, int &x, int64_t &bh, unsigned id = blockIdx.x) {
    if (xcd_map) { // nbh % 8 == 0
        const unsigned xcd = id & 7u, slot = id >> 3;
        x = (int)(slot % (unsigned)nx);
        bh = (int64_t)(slot / (unsigned)nx) * 8 + xcd;
    } else {
        x = (int)(id % (unsigned)nx);
        bh = id / (unsigned)nx;
    }
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// Mutation build (-DKF_MUTANT, kfunca_amd/_build.py: build_mutant; never in libkfunca_hip.so): deliberate single-tile defects, switched on at
// run time by kfmut_select(), that the parity tests must REJECT (tests/test_gpu_attention_mutants.py) - the proof that their bounds can fail.
//   1  forward: the last 256-query block of every head skips its key tile 1 (64 keys dropped from O and LSE)
//   2  dK/dV: the fifth 32-query slice (queries 128..159, the first one below the first key block's diagonal) contributes nothing to
//      that 128-key block (P = 0 there: dK, dV of those keys and, through the stored dS, dQ of those queries lose one slice x block).
//      (The LAST slice instead is a defect of ~1 % of those rows' norms - softmax weights of ~1/4096 beside ~1/n - which no bound
//      that tolerates 16-bit rounding can see: the bounds' resolution is a few eps of a row's norm.)
//   3  dQ (stored-dS form): the last 256-query block skips its key step 0 (64 keys dropped from dQ)
//   4  dQ (recomputing form): the same
#ifdef KF_MUTANT
#define KF_MUT(N, COND) (a.mutant == (N) && (COND))
#else
#define KF_MUT(N, COND) false
#endif

// ==========================================================================================
// MFMA path, D = 128 | 64 (template parameter; the LDS images are those of D = 128 for both)
// ==========================================================================================
constexpr int AD = 128;              // head size the LDS images are laid out for
constexpr float kPShiftF16 = 16384.f; // f16 backward: P is carried as P 2^14 into the dV product (see K4_CVTP; tools/gen_attn_dkv.py P_SHIFT)
constexpr int AROW = AD * 2;         // bytes per row of a 16-bit tile
constexpr int ABQ = 128, ABK = 64;   // forward / dQ: queries per block, keys per tile
constexpr int OPAD = AROW + 8;       // epilogue staging row stride (bytes)

template <bool BF> struct AFrag { using type = f16x8; };
template <> struct AFrag<true> { using type = bf16x8; };

template <bool BF>
__device__ __forceinline__ f32x16 a_mfma(typename AFrag<BF>::type a, typename AFrag<BF>::type b, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// LDS image of a [rows][128 x 16-bit] tile with 256-B rows: 16-B chunk `ch` of row `row` lives at
// chunk position ch ^ (((row & 3) << 2) | ((row >> 2) & 3)). Conflict-free for ds_read_b128 row
// reads of 16 consecutive rows and for the 4-row x 16-column blocks of ds_read_b64_tr_b16.
__device__ __forceinline__ int a_off(int row, int ch) { return row * AROW + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }

// A/B fragment of row `row`, k-chunk `ch` (8 consecutive elements)

// A fragment of T^T for the product  T^T(cols col0..col0+31) x X  where X is an MFMA accumulator
// used as the B operand (k order: element j of lane half h <-> row r0 + 8*(j>>2) + 4*h + (j&3)).
// T is the row-major LDS tile; two transposed reads fetch rows r0+4h+{0..3} and r0+8+4h+{0..3}.

// registers 8s..8s+7 of an accumulator -> 16-bit B fragment of k-step s
template <bool BF>
__device__ __forceinline__ typename AFrag<BF>::type a_pack(const f32x16 &x, int s) {
    typename AFrag<BF>::type r;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if constexpr (BF)
            r[j] = (__bf16)x[8 * s + j];
        else
            r[j] = (_Float16)x[8 * s + j];
    }
    return r;
}

// accumulator row index of register e for lane half h
__device__ __forceinline__ int a_row(int e, int h) { return (e & 3) + 8 * (e >> 2) + 4 * h; }

// global [nrows][128] 16-bit rows -> registers -> swizzled LDS tile, coalesced 256-B rows.
// Named members, not arrays: hipcc keeps by-reference uint4[N] staging arrays in scratch.
struct Stage4 { uint4 a, b, c, d; }; // 64 rows
struct Stage2 { uint4 a, b; };       // 32 rows
// global [nrows][128] 16-bit rows -> swizzled LDS tile, 16 bytes per call
__device__ __forceinline__ uint4 a_gld(const char *g, int id, int64_t rs) { return *(const uint4 *)(g + (int64_t)(id >> 4) * rs + (id & 15) * 16); }
__device__ __forceinline__ void a_lst(char *tile, int id, const uint4 &v) { *(uint4 *)(tile + a_off(id >> 4, id & 15)) = v; }

template <bool BF>
__device__ __forceinline__ uint32_t a_cvt16(float v) { return BF ? f32_to_bf16(v).x : f32_to_f16(v).x; }

// Write a wave's 32 x D result held as X^T accumulators (lane = row, registers = columns of
// DB = D / 32 column blocks of 32) as 16-bit rows of `dst` (row stride rs), via a per-wave LDS slab.
template <bool BF, int DB, int N>
__device__ __forceinline__ void a_store_rows(char *slab, char *dst, const f32x16 (&acc)[N], float mul, int64_t rs, int nrows = 32) {
    static_assert(DB <= N, "column blocks");
    const int lane = threadIdx.x & 63, xl = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int db = 0; db < DB; ++db)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const uint32_t h0 = a_cvt16<BF>(acc[db][4 * gq + 0] * mul), h1 = a_cvt16<BF>(acc[db][4 * gq + 1] * mul);
            const uint32_t h2 = a_cvt16<BF>(acc[db][4 * gq + 2] * mul), h3 = a_cvt16<BF>(acc[db][4 * gq + 3] * mul);
            uint2 w;
            w.x = h0 | (h1 << 16);
            w.y = h2 | (h3 << 16);
            const int col = db * 32 + 8 * gq + 4 * hl;
            *(uint2 *)(slab + xl * OPAD + col * 2) = w;
        }
    // same wave reads back what it wrote: LDS ops of one wave complete in order
#pragma unroll
    for (int i = 0; i < 4 * DB; ++i) {
        const int id = lane + 64 * i; // 256 DB pieces of 8 B: 32 rows x 8 DB pieces
        const int row = id / (8 * DB), piece = id % (8 * DB);
        const uint2 w = *(const uint2 *)(slab + row * OPAD + piece * 8);
        if (row < nrows) *(uint2 *)(dst + (int64_t)row * rs + piece * 8) = w; // (nrows < 32: the tensor's last rows, a ragged sequence length)
    }
}

// ------------------------------------------------------------------------------------------
// forward / dQ skeleton: 8 waves x 32 query rows (256-row Q block, two waves per SIMD), K/V tiles of 64 keys in an
// LDS ring filled by LDS-DMA, mask code only on the diagonal tile, scale folded into the exponent's FMA, every LDS
// address a loop-invariant VGPR + immediate.
// ------------------------------------------------------------------------------------------
constexpr int FQ = 256;                 // queries per block
constexpr int FNT = 512;                // threads per block
constexpr int FTILE = ABK * AROW;       // bytes of one K or V tile (16 KiB)
constexpr int FBUF = 2 * FTILE;         // one ring slot: K tile | V tile
constexpr int FRING = 3;                // ring depth (tiles t, t+1, t+2)

// per-lane byte offset (relative to the tile, for a 16-row-aligned r0) of the two transposed reads
__device__ __forceinline__ int a_tr_lane_off(int col0, int second) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, i = lane & 15, qq = i >> 2, p = i & 3, h = g >> 1;
    const int ch = ((col0 + 16 * (g & 1)) >> 3) + (p >> 1);
    return a_off(4 * h + qq + 8 * second, ch) + 8 * (p & 1);
}

template <bool BF>
__device__ __forceinline__ typename AFrag<BF>::type a_tr_frag2(const char *p0, const char *p1) {
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4 *)p1);
    s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename AFrag<BF>::type, r);
}

// Transposed reads issued from inline asm (the ds_read_tr builtin carries no memory operand, so beside an
// LDS-DMA in flight hipcc guards it with s_waitcnt vmcnt(0) and drains the ring). Form (ii) of guide §5.7:
// tr4_issue starts the eight reads of four fragments (column blocks d = 0..3 of one 16-row k-step) and
// returns at once; tr4_wait is the matching lgkmcnt(0), naming every destination so no consumer can be
// scheduled above it. ROFF = (first row of the k-step) * 256, a literal.
template <int DB> struct TrN { s16x4 lo[DB], hi[DB]; };
using Tr4 = TrN<4>;
using Tr2 = TrN<2>; // head size 64: two column blocks
template <int ROFF>
__device__ __forceinline__ void tr4_issue(const char *tile, const int (&vo)[2][2], Tr2 &t) {
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)tile;
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c8\n\tds_read_b64_tr_b16 %1, %5 offset:%c8\n\t"
                 "ds_read_b64_tr_b16 %2, %6 offset:%c8\n\tds_read_b64_tr_b16 %3, %7 offset:%c8"
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1])
                 : "v"(base + vo[0][0]), "v"(base + vo[0][1]), "v"(base + vo[1][0]), "v"(base + vo[1][1]), "i"(ROFF)
                 : "memory");
}
__device__ __forceinline__ void tr4_wait1(Tr2 &a) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]) : : "memory");
}
template <int ROFF>
__device__ __forceinline__ void tr4_issue(const char *tile, const int (&vo)[4][2], Tr4 &t) {
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)tile;
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%c16\n\tds_read_b64_tr_b16 %1, %9 offset:%c16\n\t"
                 "ds_read_b64_tr_b16 %2, %10 offset:%c16\n\tds_read_b64_tr_b16 %3, %11 offset:%c16\n\t"
                 "ds_read_b64_tr_b16 %4, %12 offset:%c16\n\tds_read_b64_tr_b16 %5, %13 offset:%c16\n\t"
                 "ds_read_b64_tr_b16 %6, %14 offset:%c16\n\tds_read_b64_tr_b16 %7, %15 offset:%c16"
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1]), "=&v"(t.lo[2]), "=&v"(t.hi[2]), "=&v"(t.lo[3]), "=&v"(t.hi[3])
                 : "v"(base + vo[0][0]), "v"(base + vo[0][1]), "v"(base + vo[1][0]), "v"(base + vo[1][1]), "v"(base + vo[2][0]),
                   "v"(base + vo[2][1]), "v"(base + vo[3][0]), "v"(base + vo[3][1]), "i"(ROFF)
                 : "memory");
}
__device__ __forceinline__ void tr4_wait1(Tr4 &a) {
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]), "+v"(a.lo[2]), "+v"(a.hi[2]), "+v"(a.lo[3]), "+v"(a.hi[3])
                 :
                 : "memory");
}
// the same wait with the NEXT group (one tr4_issue = 2 DB reads) still in flight
__device__ __forceinline__ void tr4_wait_next(Tr2 &a) {
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]) : : "memory");
}
__device__ __forceinline__ void tr4_wait_next(Tr4 &a) {
    asm volatile("s_waitcnt lgkmcnt(8)"
                 : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]), "+v"(a.lo[2]), "+v"(a.hi[2]), "+v"(a.lo[3]), "+v"(a.hi[3])
                 :
                 : "memory");
}
template <bool BF, int DB>
__device__ __forceinline__ typename AFrag<BF>::type tr4_frag(const TrN<DB> &t, int d) {
    s16x8 r;
    r[0] = t.lo[d][0]; r[1] = t.lo[d][1]; r[2] = t.lo[d][2]; r[3] = t.lo[d][3];
    r[4] = t.hi[d][0]; r[5] = t.hi[d][1]; r[6] = t.hi[d][2]; r[7] = t.hi[d][3];
    return __builtin_bit_cast(typename AFrag<BF>::type, r);
}


// LDS-DMA staging of one 64-key K tile and V tile (global_load_lds_dwordx4): 16 + 16 wave-instructions of
// 1 KiB (4 rows), two of each per wave; the tile image's XOR swizzle goes on the per-lane SOURCE chunk.
// Head size 64 keeps the 256-byte row pitch of the image (every offset below stays what it is); its rows hold 8 chunks, so the
// lanes whose source chunk is 8..15 sit the instruction out (an LDS-DMA lane writes at base + 16 lane whatever the others do).
// (Round 3, tried and dropped: the BUFFER form - buffer_load_dwordx4 ... offen lds with a per-tile scalar descriptor and a loop-invariant
// 32-bit lane offset instead of a 64-bit address per lane: the forward ran 1.58-1.61 ms against 1.53-1.56 on the same box. The ~108
// cycles an LDS-DMA instruction holds a wave's issue here are not address arithmetic: all eight waves issue their four pieces right
// behind the tile barrier, 32 KiB in one burst through a 64 B / clock vector-memory path.)
template <int D>
__device__ __forceinline__ void f_stage(const char *kg, const char *vg, char *buf, int64_t krs, int64_t vrs) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row0 = (wid * 2 + i) * 4, row = row0 + (lane >> 4), pos = lane & 15;
        const int chunk = pos ^ (((row & 3) << 2) | ((row >> 2) & 3));
        if (D == AD || chunk < D / 8) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kg + row * krs + chunk * 16),
                                             (__attribute__((address_space(3))) void *)(buf + row0 * AROW), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(vg + row * vrs + chunk * 16),
                                             (__attribute__((address_space(3))) void *)(buf + FTILE + row0 * AROW), 16, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------
// forward: the two waves of each SIMD are STAGGERED. A compile-time ablation of the unstaggered version of this kernel
// (removed; numbers in DESIGN.md) showed its QK, softmax and PV costs add up (0.37 + 0.55 + 0.47 ms of 1.96): all eight
// waves leave the tile barrier
// together, so both waves of a SIMD sit in the same phase and the matrix pipe idles while both do
// softmax arithmetic. Here waves 4-7 run one phase late: in tile interval t they do PV(t-1), QK(t),
// softmax(t) while waves 0-3 do QK(t), softmax(t), PV(t) — two of the three phases pair matrix work
// with vector work. Costs one more ring slot (V of tile t-1 must survive interval t).
// ------------------------------------------------------------------------------------------
// combine a value with the one held by the lane 32 away: ONE v_permlane32_swap (VALU) instead of a ds_bpermute round trip
// through the LDS crossbar. swap(x, x) leaves {lower half's x in all lanes, upper half's x in all lanes}.
__device__ __forceinline__ float a_half_max(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float a_half_sum(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}

constexpr int SRING = 4;
constexpr float kDeferMax = 8.0f;

// Diagnostic build only (-DKF_ATTN_TIMELINE, tools/attn_timeline.py; never in libkfunca_hip.so): every wave of the forward adds up the
// shader-clock cycles it spends in each phase of its tile loop (s_memtime stamps at the phase boundaries) and writes the seven sums.
#ifdef KF_ATTN_TIMELINE
__device__ unsigned long long *g_attn_tl;
#define TL_STAMP(I)                                                        \
    {                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                 \
        const unsigned long long n_ = __builtin_amdgcn_s_memtime();        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 \
        tl_acc[I] += n_ - tl_t;                                            \
        tl_t = n_;                                                         \
        __builtin_amdgcn_sched_barrier(0);                                 \
    }
// dK/dV slice: the stamps are only REQUESTED at the phase boundaries (a wait would drain the LDS prefetch the slice lives on) and
// collected at the end of the slice; the counted LDS waits of the slice see up to ten more outstanding operations and get that much
// more conservative.
#define TLK_STAMP(I) asm volatile("s_memtime %0" : "=s"(tlk[I]) : : "memory");
#define TL_PARAMS , unsigned long long &tl_t, unsigned long long (&tl_acc)[7]
#define TL_ARGS , tl_t, tl_acc
#else
#define TL_STAMP(I)
#define TLK_STAMP(I)
#define TL_PARAMS
#define TL_ARGS
#endif

template <bool BF, bool MASK, int D>
__device__ __forceinline__ void s_qk_sm(const char *buf, const typename AFrag<BF>::type (&qf)[D / 16], const int (&ko)[D / 16], f32x16 (&o)[D / 32],
                                        typename AFrag<BF>::type (&pf)[4], float &m_i, float &l_i, float c, int64_t kv0, int64_t m, int hl, float defer TL_PARAMS) {
    using frag_t = typename AFrag<BF>::type;
    f32x16 s[2];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[sub][e] = 0.f;
#pragma unroll
        for (int kg = 0; kg < D / 64; ++kg) {
#pragma unroll
            for (int kk = 4 * kg; kk < 4 * kg + 4; ++kk)
                s[sub] = a_mfma<BF>(*(const frag_t *)(buf + sub * 32 * AROW + ko[kk]), qf[kk], s[sub]);
            __builtin_amdgcn_sched_barrier(0); // bounds the K fragments in flight (register budget)
        }
    }
    TL_STAMP(3)
    float mx = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (MASK && kv0 + sub * 32 + a_row(e, hl) > m) s[sub][e] = -INFINITY;
            mx = fmaxf(mx, s[sub][e]);
        }
    mx = a_half_max(mx);
    // deferred running maximum: m_i (raw score units) is the maximum IN USE; a larger one is adopted - and O, l rescaled: 66
    // multiplies - only when some query of the wave exceeds it by more than kDeferMax exponent units. Until then p can reach
    // 2^kDeferMax, which neither the 16-bit P nor the f32 row sum minds; after the first few tiles the branch is rarely taken.
    if (__builtin_amdgcn_ballot_w64((mx - m_i) * c > defer) != 0) {
        const float m_new = fmaxf(m_i, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_i - m_new) * c);
        l_i *= alpha;
#pragma unroll
        for (int d = 0; d < D / 32; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        m_i = m_new;
    }
    const float mc = m_i * c;
    float rs = 0.f;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[sub][e], c, -mc));
            s[sub][e] = p;
            rs += p;
        }
    rs = a_half_sum(rs);
    l_i += rs;
    pf[0] = a_pack<BF>(s[0], 0);
    pf[1] = a_pack<BF>(s[0], 1);
    pf[2] = a_pack<BF>(s[1], 0);
    pf[3] = a_pack<BF>(s[1], 1);
}

template <bool BF, int DB>
__device__ __forceinline__ void s_pv(const char *vt, const int (&vo)[DB][2], const typename AFrag<BF>::type (&pf)[4], f32x16 (&o)[DB]) {
    // two groups of transposed V fragments: group g + 1 is in flight under the MFMAs of group g (round 3; one group at a time, with the
    // partner wave covering the LDS round trips, measured 1.38-1.41 ms against 1.36-1.38 for this form on the same box: a small gain)
    TrN<DB> ta, tb;
    tr4_issue<0>(vt, vo, ta);
    tr4_issue<16 * AROW>(vt, vo, tb);
    tr4_wait_next(ta);
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = a_mfma<BF>(tr4_frag<BF, DB>(ta, d), pf[0], o[d]);
    __builtin_amdgcn_sched_barrier(0);
    tr4_issue<32 * AROW>(vt, vo, ta);
    tr4_wait_next(tb);
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = a_mfma<BF>(tr4_frag<BF, DB>(tb, d), pf[1], o[d]);
    __builtin_amdgcn_sched_barrier(0);
    tr4_issue<48 * AROW>(vt, vo, tb);
    tr4_wait_next(ta);
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = a_mfma<BF>(tr4_frag<BF, DB>(ta, d), pf[2], o[d]);
    __builtin_amdgcn_sched_barrier(0);
    tr4_wait1(tb);
#pragma unroll
    for (int d = 0; d < DB; ++d) o[d] = a_mfma<BF>(tr4_frag<BF, DB>(tb, d), pf[3], o[d]);
    __builtin_amdgcn_sched_barrier(0);
}

template <bool BF, int D>
__global__ __launch_bounds__(FNT, 2) void attn_fwd_v3_kernel(const AttnArgs a) {
    using frag_t = typename AFrag<BF>::type;
    constexpr int KS = D / 16, DB = D / 32; // k-steps of a Q K^T chain, 32-wide column blocks of O
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, xl = lane & 31, hl = lane >> 5;
    int xb0;
    int64_t bh;
    const int nxb = (int)((a.Sq + FQ - 1) / FQ);
    // a.persist (nxb even): a workgroup takes query block x AND its mirror nxb - 1 - x of the same (batch, head), so every
    // workgroup has the same causal work (nxb + 1 key tiles' worth) and half as many workgroups are dispatched
    const int nwx = a.persist ? nxb / (2 * a.persist) : nxb; // workgroups per (batch, head)
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    const int rgrp = ((wid & 3) << 1) | (wid >> 2);
    const bool late = __builtin_amdgcn_readfirstlane(wid) >= 4;
    const char *Kg = a.k + a_head(a.lk, bh, a.H);
    const char *Vg = a.v + a_head(a.lv, bh, a.H);
    int ko[KS], vo[DB][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ko[kk] = a_off(xl, kk * 2 + hl);
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        vo[d][0] = a_tr_lane_off(d * 32, 0);
        vo[d][1] = a_tr_lane_off(d * 32, 1);
    }
    const float c = a.scale * kLog2e;
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;           // pair index: blocks xp and nxb - 1 - xp
    const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nxb - 1 - xp : xp;
    const int qblk = nxb - 1 - xb; // longest blocks first
#ifdef KF_ATTN_TIMELINE
    const unsigned long long tl_pass0 = __builtin_amdgcn_s_memtime();
#endif
    // query rows are dealt so that each SIMD's early wave (w) and late wave (w + 4) own ADJACENT 32-row groups:
    // their causal work differs by at most one tile
    const int64_t q0 = (int64_t)qblk * FQ, qw = q0 + rgrp * 32, m = qw + xl;
    const bool active = qw < a.Sq;

    frag_t qf[KS];
    if (active) {
        const char *Qg = a.q + a_head(a.lq, bh, a.H) + m * a.lq.sr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[kk] = *(const frag_t *)(Qg + (kk * 16 + 8 * hl) * 2);
    } else {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = 0;
    }
    f32x16 o[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    frag_t pf[4];
    bool pending = false; // late waves: P of the previous tile still waits for its PV

    const int64_t q_end = q0 + FQ < a.Sq ? q0 + FQ : a.Sq;
    const int64_t kv_end = a.Skv < q_end ? a.Skv : q_end;
    const int nt = (int)((kv_end + ABK - 1) / ABK);
    auto stage = [&](int tile, char *buf) {
        const int64_t kv = (int64_t)(tile < nt ? tile : nt - 1) * ABK;
        f_stage<D>(Kg + kv * a.lk.sr, Vg + kv * a.lv.sr, buf, a.lk.sr, a.lv.sr);
    };
    stage(0, smem);
    stage(1, smem + FBUF);
#ifdef KF_ATTN_TIMELINE
    unsigned long long tl_t = __builtin_amdgcn_s_memtime(), tl_acc[7] = {0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tl_loop0 = tl_t;
#endif
    for (int t = 0; t < nt; ++t) {
        const int64_t kv0 = (int64_t)t * ABK;
        TL_STAMP(6) // loop overhead
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        TL_STAMP(0) // this wave's part of tile t has landed
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        TL_STAMP(1) // everyone's
        // (round 3, tried: a five-slot ring (all 160 KiB) with ONE barrier per TWO tiles - 1.54 ms against 1.41-1.47 on the same box: the
        // per-tile barrier is what keeps the early / late waves of a SIMD in their complementary phases.)
        // (round 3, tried: the four LDS-DMA instructions under the first four MFMAs of the score chain instead of here in the open, where
        // they cost the wave 433 cycles per tile - 1.41 ms against 1.28 on the same box and run: with two waves per SIMD a wave stuck
        // issuing DMA costs the SIMD nothing, its partner has the pipes, while the same instructions inside the chain delay the chain)
        stage(t + 2, smem + ((t + 2) % SRING) * FBUF); // slot of tile t-2: nobody reads it any more
        TL_STAMP(2) // four LDS-DMA instructions issued
        const char *cur = smem + (t % SRING) * FBUF;
        const bool skip = !active || kv0 > qw + 31 || KF_MUT(1, t == 1 && qblk == nxb - 1);
        const bool diag = kv0 + ABK - 1 > qw;
        // one copy of each phase in program order [PV(t-1) | QK+softmax(t) | PV(t)]: late waves take the first
        // two, early waves the last two
        if (late && pending) s_pv<BF, DB>(smem + ((t + SRING - 1) % SRING) * FBUF + FTILE, vo, pf, o);
        pending = false;
        TL_STAMP(5) // a late wave's P V of the previous tile
        if (!skip) {
            if (diag) s_qk_sm<BF, true, D>(cur, qf, ko, o, pf, m_i, l_i, c, kv0, m, hl, a.defer TL_ARGS); // stamps 3 after Q K^T
            else s_qk_sm<BF, false, D>(cur, qf, ko, o, pf, m_i, l_i, c, kv0, m, hl, a.defer TL_ARGS);
            TL_STAMP(4) // softmax
            if (late) pending = true;
            else s_pv<BF, DB>(cur + FTILE, vo, pf, o);
            TL_STAMP(5) // an early wave's P V
        }
    }
#if defined(KF_ATTN_TIMELINE) && KF_ATTN_TIMELINE != 2
    if (lane == 0 && g_attn_tl) {
        unsigned long long *dst = g_attn_tl + ((size_t)blockIdx.x * 2 + (pass & 1)) * 64 + wid * 8;
        for (int i = 0; i < 7; ++i) dst[i] = tl_acc[i];
        dst[7] = (unsigned long long)nt;
    }
#endif
#ifdef KF_ATTN_TIMELINE
    const unsigned long long tl_loop1 = __builtin_amdgcn_s_memtime();
#endif
    if (late && pending) s_pv<BF, DB>(smem + ((nt + SRING - 1) % SRING) * FBUF + FTILE, vo, pf, o);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) {
        a_store_rows<BF, DB>(smem + wid * 32 * OPAD, a.out + a_head(a.lo, bh, a.H) + qw * a.lo.sr, o, 1.f / l_i, a.lo.sr);
        if (a.lse && hl == 0) a.lse[bh * a.Sq + m] = (m_i * c + __builtin_amdgcn_logf(l_i)) * kLn2;
    }
    if (a.persist) __syncthreads(); // the staging slabs overlap the ring the next block fills
#if defined(KF_ATTN_TIMELINE) && KF_ATTN_TIMELINE == 2 // per pass: before the tile loop | the loop | behind it (tools/attn_timeline.py --outside)
    if (lane == 0 && g_attn_tl) {
        unsigned long long *dst = g_attn_tl + ((size_t)blockIdx.x * 2 + (pass & 1)) * 64 + wid * 8;
        const unsigned long long tl_end = __builtin_amdgcn_s_memtime();
        dst[0] = tl_loop0 - tl_pass0, dst[1] = tl_loop1 - tl_loop0, dst[2] = tl_end - tl_loop1;
        dst[3] = dst[4] = dst[5] = dst[6] = 0;
        dst[7] = (unsigned long long)nt;
    }
#endif
  }
}

// ------------------------------------------------------------------------------------------
// forward, round 4: 4 waves x 64 query rows, ONE wave per SIMD, the whole 512-register file asm-owned, the query block's pass (prologue,
// tile loop in six variants, epilogue) as ONE generated instruction stream: tools/gen_attn_fwd.py -> attn_fwd_w4.inc (structure, register
// map and the placement tables are documented there). This wrapper only maps the workgroup to its two query blocks (a block and its
// causal mirror, as the other kernels) and hands the stream its scalars; every lane-dependent value is formed inside.
// Head size 128, Sq a multiple of 256, Skv >= Sq, K and V with the same row stride; everything else keeps attn_fwd_v3_kernel.
// ------------------------------------------------------------------------------------------
#ifndef KF_FWD_W4_INC // (tools/scratch/fwd_w4_ablate.sh builds timing variants of the stream from another file)
#define KF_FWD_W4_INC "attn_fwd_w4.inc"
#endif
#include KF_FWD_W4_INC
// SQ: the scaled-query form of the stream (KF_ATTN_SCALED_OPERANDS): c q rounded to the element type once per pass, no multiply per score -
// about 4 % faster, and a score error of eps scale sum |q k| that grows with the logits. Default: exact f32 scores.
// D64: the head-size-64 stream (round 5; exact scores only)
template <bool BF, bool SQ, bool D64 = false>
__global__ __launch_bounds__(256) void attn_fwd_w4_kernel(const AttnArgs a) {
    static_assert(!(SQ && D64), "the scaled-query form exists for head size 128 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nxb = (int)((a.Sq + FQ - 1) / FQ);
    const int nwx = a.persist ? nxb / 2 : nxb;
    const unsigned lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dbytes = D64 ? 128 : 256;
    // what the stream's descriptors may touch: K / V rows below Skv (bytes from the head's base; rows beyond read as zeros)
    const unsigned kvn = (unsigned)((a.Skv - 1) * a.lk.sr + dbytes);
    const float c = a.scale_log2e, defer = a.defer; // (kernel arguments are scalar registers; a float product formed here would be a vector one)
    const int qsr = (int)a.lq.sr, kvsr = (int)a.lk.sr, osr = (int)a.lo.sr;
    // (a.nvwg virtual workgroups over gridDim.x real ones: KF_ATTN_GRID_WGS, an experiment - default one each. A stride of a multiple of 8 keeps
    //  a virtual workgroup on the XCD its id maps to.)
#pragma nounroll
    for (unsigned vwg = blockIdx.x; vwg < a.nvwg; vwg += gridDim.x) {
    int xb0;
    int64_t bh;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh, vwg);
    const char *kp = a.k + a_head(a.lk, bh, a.H), *vp = a.v + a_head(a.lv, bh, a.H);
    const char *qh = a.q + a_head(a.lq, bh, a.H);
    char *oh = a.out + a_head(a.lo, bh, a.H);
#pragma nounroll
    for (int pass = 0; pass < (a.persist ? 2 : 1); ++pass) {
        const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nxb - 1 - xb0 : xb0;
        const int qblk = nxb - 1 - xb; // longest blocks first
        const int64_t q0 = (int64_t)qblk * FQ;
        const char *qp = qh + q0 * a.lq.sr;
        char *op = oh + q0 * a.lo.sr;
        float *lsep = a.lse ? a.lse + bh * a.Sq + q0 : nullptr;
        const int T = (int)((q0 + FQ) / ABK); // key tiles of this block: up to its last query's diagonal (tiles beyond Skv arrive as zeros)
        const int64_t qrows = a.Sq - q0 < FQ ? a.Sq - q0 : FQ; // the block's rows that exist: Q rows beyond them read as zeros, O / lse rows beyond are not stored
        const unsigned qn = (unsigned)((qrows - 1) * a.lq.sr + dbytes), on = (unsigned)((qrows - 1) * a.lo.sr + dbytes), lsen = (unsigned)(qrows * 4);
        int mut = -1;
#ifdef KF_MUTANT
        if (a.mutant == 1 && qblk == nxb - 1) mut = 1; // defect 1: the head's last block drops key tile 1
#endif
#ifdef KF_FWD_W4_STAMPS // diagnostic build (tools/attn_fwd_w4_timeline.py): eight cycle sums per wave and block into the debug buffer
        unsigned long long *dbg = a.dbg;
        const unsigned dbgoff = (vwg * 2 + pass) * 4 * 32;
#define KF_W4_EXTRA , [dbg] "s"(dbg), [dbgoff] "s"(dbgoff)
#else
#define KF_W4_EXTRA
#endif
#define KF_W4_OPERANDS                                                                                                              \
    [qp] "s"(qp), [kp] "s"(kp), [vp] "s"(vp), [op] "s"(op), [lsep] "s"(lsep), [qsr] "s"(qsr), [kvsr] "s"(kvsr), [osr] "s"(osr), [T] "s"(T), \
        [wid] "s"(wid), [c] "s"(c), [defer] "s"(defer), [lds] "s"(lds), [mut] "s"(mut), [kvn] "s"(kvn), [qn] "s"(qn), [on] "s"(on), [lsen] "s"(lsen) KF_W4_EXTRA
        if constexpr (D64 && BF) asm volatile(KF_FWD_W4_D64_ASM_BF16 : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
        else if constexpr (D64) asm volatile(KF_FWD_W4_D64_ASM_F16 : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
        else if constexpr (BF && SQ) asm volatile(KF_FWD_W4_ASM_BF16_SQ : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
        else if constexpr (BF) asm volatile(KF_FWD_W4_ASM_BF16 : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
        else if constexpr (SQ) asm volatile(KF_FWD_W4_ASM_F16_SQ : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
        else asm volatile(KF_FWD_W4_ASM_F16 : : KF_W4_OPERANDS : KF_FWD_W4_CLOBBERS);
#undef KF_W4_OPERANDS
#undef KF_W4_EXTRA
    }
    }
}

// ------------------------------------------------------------------------------------------
// backward pre-pass: delta[q] = sum_d dO[q][d] * O[q][d]   (16 lanes per row, 16-B loads)
// ------------------------------------------------------------------------------------------
template <bool BF, int R>
__global__ __launch_bounds__(256) void attn_delta_kernel(const char *o, const char *d_o, float *delta, int64_t nrows, const float *lse, float *nlse,
                                                         float *ndelta, float rscale, AttnArgs::Lay lo, AttnArgs::Lay ldo, int64_t S, int64_t H, int nparts, int64_t Sc) {
    // 16 lanes per row, R rows per 16-lane group (rows row0 + 16 i): 2 R 16-byte loads in flight per lane
    // rows are numbered over [B H, Sc], Sc = S rounded up to 32: delta stays [B, H, S] contiguous, the two row-constant arrays are [B H, Sc] with
    // ZERO pad rows (a ragged last slice of the dK/dV kernels reads its 32 constants: a zero constant beside a zero Q / dO row is harmless)
    const int64_t row0 = (int64_t)blockIdx.x * (16 * R) + (threadIdx.x >> 4);
    const int part = threadIdx.x & 15;
    uint4 a[R], b[R];
#pragma unroll
    for (int i = 0; i < R; ++i) {
        a[i] = b[i] = uint4{0, 0, 0, 0};
        const int64_t row = row0 + 16 * i;
        const int64_t bh = row / Sc, sq = row - bh * Sc;
        if (row < nrows && sq < S && part < nparts) { // nparts = D / 8 (16 | 8): 16-byte pieces of a row
            a[i] = *(const uint4 *)(o + a_head(lo, bh, H) + sq * lo.sr + part * 16);
            b[i] = *(const uint4 *)(d_o + a_head(ldo, bh, H) + sq * ldo.sr + part * 16);
        }
    }
#pragma unroll
    for (int i = 0; i < R; ++i) {
        const int64_t row = row0 + 16 * i;
        float acc = 0.f;
        const uint32_t aw[4] = {a[i].x, a[i].y, a[i].z, a[i].w}, bw[4] = {b[i].x, b[i].y, b[i].z, b[i].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a0, a1, b0, b1;
            if constexpr (BF) {
                a0 = __uint_as_float(aw[j] << 16); a1 = __uint_as_float(aw[j] & 0xffff0000u);
                b0 = __uint_as_float(bw[j] << 16); b1 = __uint_as_float(bw[j] & 0xffff0000u);
            } else {
                a0 = f16_to_f32(f16_t{(uint16_t)(aw[j] & 0xffff)}); a1 = f16_to_f32(f16_t{(uint16_t)(aw[j] >> 16)});
                b0 = f16_to_f32(f16_t{(uint16_t)(bw[j] & 0xffff)}); b1 = f16_to_f32(f16_t{(uint16_t)(bw[j] >> 16)});
            }
            acc += a0 * b0 + a1 * b1;
        }
        for (int msk = 8; msk > 0; msk >>= 1) acc += __shfl_xor(acc, msk, 64);
        if (row < nrows && part == 0) {
            const int64_t bh = row / Sc, sq = row - bh * Sc;
            if (sq < S) {
                delta[bh * S + sq] = acc;
                if (nlse) { // row constants of the dK/dV kernels: S' = Q K^T - lse / scale and dP' = dO V^T - delta come out of the MFMA chains ready
                    nlse[row] = -lse[bh * S + sq] * rscale;
                    ndelta[row] = -acc;
                }
            } else if (nlse) {
                nlse[row] = 0.f;
                ndelta[row] = 0.f;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// backward: dQ, v2 — the forward-v2 skeleton (8 waves x 32 queries, two waves per SIMD, K/V tiles
// double-buffered, one barrier per tile). Per 32-key sub-tile: S^T = K Q^T, dP^T = V dO^T,
// dS^T = P^T o (dP^T - delta), dQ^T += K^T dS^T (K^T through transposed reads of the same K image).
// ------------------------------------------------------------------------------------------
template <bool BF, bool MASK, int D>
__device__ __forceinline__ void q_tile(const char *buf, const char *doslab, const typename AFrag<BF>::type (&qf)[D / 16], const int (&ko)[D / 16],
                                       const int (&vo)[D / 32][2], f32x16 (&dq)[D / 32], float c, float lse2, float dlt, int64_t kv0, int64_t m, int hl) {
    using frag_t = typename AFrag<BF>::type;
    constexpr int DB = D / 32;
    const char *vt = buf + FTILE;
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
        f32x16 s, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = 0.f; dp[e] = 0.f; }
        // groups of four k-steps fenced for the scheduler: hoisting every fragment load of the tile to the
        // top costs > 90 VGPRs and spills at two waves per SIMD; the partner wave hides the LDS latency instead
#pragma unroll
        for (int kg = 0; kg < D / 64; ++kg) {
#pragma unroll
            for (int kk = 4 * kg; kk < 4 * kg + 4; ++kk) s = a_mfma<BF>(*(const frag_t *)(buf + sub * 32 * AROW + ko[kk]), qf[kk], s);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int kg = 0; kg < D / 64; ++kg) {
#pragma unroll
            for (int kk = 4 * kg; kk < 4 * kg + 4; ++kk)
                dp = a_mfma<BF>(*(const frag_t *)(vt + sub * 32 * AROW + ko[kk]), *(const frag_t *)(doslab + ko[kk]), dp);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[e], c, -lse2));
            if (MASK && kv0 + sub * 32 + a_row(e, hl) > m) p = 0.f;
            s[e] = p * (dp[e] - dlt);
        }
        TrN<DB> ta; // one group of transposed K fragments at a time (register budget); the partner wave covers the LDS latency
        if (sub == 0) tr4_issue<0>(buf, vo, ta); else tr4_issue<32 * AROW>(buf, vo, ta);
        tr4_wait1(ta);
        { const frag_t df = a_pack<BF>(s, 0);
#pragma unroll
          for (int d = 0; d < DB; ++d) dq[d] = a_mfma<BF>(tr4_frag<BF, DB>(ta, d), df, dq[d]); }
        __builtin_amdgcn_sched_barrier(0);
        if (sub == 0) tr4_issue<16 * AROW>(buf, vo, ta); else tr4_issue<48 * AROW>(buf, vo, ta);
        tr4_wait1(ta);
        { const frag_t df = a_pack<BF>(s, 1);
#pragma unroll
          for (int d = 0; d < DB; ++d) dq[d] = a_mfma<BF>(tr4_frag<BF, DB>(ta, d), df, dq[d]); }
        __builtin_amdgcn_sched_barrier(0);
    }
}

constexpr int QSLAB = 32 * AROW;                 // one wave's dO rows (8 KiB)
constexpr int QLDS = FRING * FBUF + 8 * QSLAB;   // K/V ring + 8 dO slabs = 160 KiB (the whole LDS of a CU)

template <bool BF, int D>
__global__ __launch_bounds__(FNT, 2) void attn_bwd_dq_v2_kernel(const AttnArgs a) {
    using frag_t = typename AFrag<BF>::type;
    constexpr int KS = D / 16, DB = D / 32; // head size 64: half the k-steps and column blocks, the LDS images keep their 256-byte rows
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, xl = lane & 31, hl = lane >> 5;
    int xb0;
    int64_t bh;
    const int nxb = (int)((a.Sq + FQ - 1) / FQ);
    // a.persist: a workgroup takes query block x and its causal mirror nxb - 1 - x (equal work per workgroup, as in the forward)
    const int nwx = a.persist ? nxb / (2 * a.persist) : nxb;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    const char *Kg = a.k + a_head(a.lk, bh, a.H);
    const char *Vg = a.v + a_head(a.lv, bh, a.H);
    char *doslab = smem + FRING * FBUF + wid * QSLAB; // this wave's dO rows, same swizzled image as a K tile (B operand of dP^T)
    int ko[KS], vo[DB][2];
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) ko[kk] = a_off(xl, kk * 2 + hl);
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        vo[d][0] = a_tr_lane_off(d * 32, 0);
        vo[d][1] = a_tr_lane_off(d * 32, 1);
    }
    const float c = a.scale * kLog2e;
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nxb - 1 - xp : xp;
    const int qblk = nxb - 1 - xb; // longest blocks first
    const int64_t q0 = (int64_t)qblk * FQ, qw = q0 + wid * 32, m = qw + xl;
    const bool active = qw < a.Sq;

    frag_t qf[KS];
    float lse2 = 0.f, dlt = 0.f;
    if (active) {
        const char *Qg = a.q + a_head(a.lq, bh, a.H) + m * a.lq.sr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) qf[kk] = *(const frag_t *)(Qg + (kk * 16 + 8 * hl) * 2);
        const char *dOw = a.d_o + a_head(a.ldo, bh, a.H) + qw * a.ldo.sr;
#pragma unroll
        for (int i = 0; i < 8; ++i) // 512 pieces of 16 B = 32 rows x 16 chunks; head size 64: chunks 8..15 lie beyond the row
            if (D == AD || ((lane + 64 * i) & 15) < D / 8) a_lst(doslab, lane + 64 * i, a_gld(dOw, lane + 64 * i, a.ldo.sr));
        lse2 = a.lse_r[bh * a.Sq + m] * kLog2e;
        dlt = a.delta[bh * a.Sq + m];
    } else {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[kk][j] = 0;
    }
    f32x16 dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;

    const int64_t q_end = q0 + FQ < a.Sq ? q0 + FQ : a.Sq;
    const int64_t kv_end = a.Skv < q_end ? a.Skv : q_end;
    const int nt = (int)((kv_end + ABK - 1) / ABK);
    auto stage = [&](int tile, char *buf) { // 3-deep ring, counted vmcnt: see attn_fwd_v2_kernel
        const int64_t kv = (int64_t)(tile < nt ? tile : nt - 1) * ABK;
        f_stage<D>(Kg + kv * a.lk.sr, Vg + kv * a.lv.sr, buf, a.lk.sr, a.lv.sr);
    };
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // Q fragments + the dO slab writes are done
    stage(0, smem);
    stage(1, smem + FBUF);
    for (int t = 0; t < nt; ++t) {
        const int64_t kv0 = (int64_t)t * ABK;
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stage(t + 2, smem + ((t + 2) % FRING) * FBUF);
        const char *cur = smem + (t % FRING) * FBUF;
        const bool skip = !active || kv0 > qw + 31 || KF_MUT(4, t == 0 && qblk == nxb - 1);
        const bool diag = kv0 + ABK - 1 > qw;
        if (!skip) {
            if (diag) q_tile<BF, true, D>(cur, doslab, qf, ko, vo, dq, c, lse2, dlt, kv0, m, hl);
            else q_tile<BF, false, D>(cur, doslab, qf, ko, vo, dq, c, lse2, dlt, kv0, m, hl);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) a_store_rows<BF, DB>(smem + wid * 32 * OPAD, a.dq + a_head(a.ldq, bh, a.H) + qw * a.ldq.sr, dq, a.scale, a.ldq.sr);
    if (a.persist) __syncthreads(); // the staging slabs overlap the ring the next block fills
  }
}

// ------------------------------------------------------------------------------------------
// backward: dQ from the stored dS (DS_TILE workspace) - dQ^T += K^T dS^T, the forward's P V half with dS in place of P and K
// in place of V: 8 waves x 32 queries (a 256-query block; wave w <-> slice w), two waves per SIMD, 64-key steps.
//   A = K^T fragments through transposed reads of the row-major K tile (the forward's V path: same image, same offsets);
//   B = dS^T fragments (k = key, column = query): each wave streams ITS slice's two 2 KiB dS tiles per step by LDS-DMA into a
//       private slab, verbatim ([s][key][hl][16 B]) but for a swap of 64-byte key pairs in the second half - that makes the
//       image conflict-free for ds_read_b64_tr_b16 (4 keys x 8 query quads of a half-wave = 32 distinct 8-byte units) while
//       every DMA instruction still reads 1 KiB of consecutive bytes; lane 16 grp + 4 qq + p reads the unit of key
//       4 h + qq, query quad 4 (grp & 1) + p.
// The K tile is shared by the workgroup (one barrier per step); the dS slabs are private, so they need no barrier at all.
// No mask and no row constants: the dK/dV kernel wrote zeros above the diagonal, and only tiles at or below a slice's
// diagonal are read (key block <= slice). HBM-bound by construction: dS is read exactly once (16 MFMAs per 4 KiB of it).
// ------------------------------------------------------------------------------------------
constexpr int DQ_RING = 3;
constexpr int DQ_SLAB = 2 * DS_TILE;                              // one wave's dS of one 64-key step
constexpr int DQ_LDS = DQ_RING * FTILE + 8 * DQ_RING * DQ_SLAB;   // K ring 48 KiB + 8 private dS rings 96 KiB

// one 16-key k-step of key block HF of the step: dS^T fragment (2 transposed reads of the private slab) + four K^T fragments
template <bool BF, int HF, int KS, int DB>
__device__ __forceinline__ void dq_step(const char *kt, const int (&vo)[DB][2], unsigned dsb, f32x16 (&dq)[DB]) {
    s16x4 blo, bhi;
    TrN<DB> ta;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4"
                 : "=&v"(blo), "=&v"(bhi)
                 : "v"(dsb), "n"(HF * DS_TILE + KS * 512), "n"(HF * DS_TILE + KS * 512 + 256)
                 : "memory");
    tr4_issue<(32 * HF + 16 * KS) * AROW>(kt, vo, ta);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(blo), "+v"(bhi) : : "memory");
    tr4_wait1(ta);
    s16x8 r;
    r[0] = blo[0]; r[1] = blo[1]; r[2] = blo[2]; r[3] = blo[3];
    r[4] = bhi[0]; r[5] = bhi[1]; r[6] = bhi[2]; r[7] = bhi[3];
    const auto b = __builtin_bit_cast(typename AFrag<BF>::type, r);
#pragma unroll
    for (int d = 0; d < DB; ++d) dq[d] = a_mfma<BF>(tr4_frag<BF, DB>(ta, d), b, dq[d]);
}

template <bool BF, int D>
__global__ __launch_bounds__(FNT, 2) void attn_bwd_dq_ds_kernel(const AttnArgs a) {
    constexpr int DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int xb0;
    int64_t bh;
    const int nxb = (int)((a.Sq + FQ - 1) / FQ);
    const int nwx = a.persist ? nxb / (2 * a.persist) : nxb;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    bh += a.bh0;
    const char *Kg = a.k + a_head(a.lk, bh, a.H);
    char *slab = smem + DQ_RING * FTILE + wid * DQ_RING * DQ_SLAB;
    int vo[DB][2];
#pragma unroll
    for (int d = 0; d < DB; ++d) {
        vo[d][0] = a_tr_lane_off(d * 32, 0);
        vo[d][1] = a_tr_lane_off(d * 32, 1);
    }
    // transposed-read address of the dS^T fragment inside a slab: key 4 h + qq, query quad 4 (grp & 1) + p
    const int grp = lane >> 4, qq = (lane >> 2) & 3, pp = lane & 3, h = grp >> 1;
    const unsigned smem_u = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    const int sp = grp & 1; // operand s' = query quads 4 s' .. 4 s' + 3
    const unsigned ds_rd = smem_u + (unsigned)(slab - smem) + sp * 1024 + 64 * ((2 * h + (qq >> 1)) ^ (2 * sp)) + 32 * (qq & 1) + 16 * (pp & 1) + 8 * (pp >> 1);
    // LDS-DMA source: a piece is one operand half s of a tile (1 KiB, stored [key][hl][16 B]); it lands verbatim except that
    // in half 1 the 64-byte key pairs a and a ^ 2 trade places (conflict-free transposed reads; every 4 lanes still read 64
    // consecutive bytes)
    int ds_src[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) ds_src[pc] = pc * 1024 + (((lane >> 2) ^ (2 * pc)) * 4 + (lane & 3)) * 16;
    const int krow = lane >> 4, kpos = lane & 15;
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nxb - 1 - xp : xp;
    const int qblk = nxb - 1 - xb; // longest blocks first
    const int64_t q0 = (int64_t)qblk * FQ, qw = q0 + wid * 32;
    const bool active = qw < a.Sq;
    const int sl = (int)(qw >> 5); // this wave's slice: key blocks 0 .. sl contribute
    f32x16 dq[DB];
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
    const int64_t q_end = q0 + FQ < a.Sq ? q0 + FQ : a.Sq;
    const int64_t kv_end = a.Skv < q_end ? a.Skv : q_end;
    const int nt = (int)((kv_end + ABK - 1) / ABK);
    // this query block's row of the pair's tiles, this wave's slice of it (whether it has queries or not: the grid holds whole query blocks): + kwb * 8 tiles
    const char *dsg = a.ds + ((bh - a.bh0) * a.ds_pair + ds_row_base(qblk, a.ds_nkb, a.ds_tri) + wid) * DS_TILE;
    const int kwb_last = (int)(a.ds_nkb * 8 - 1);
    const int slw = qblk * 8 + wid;
    auto stage = [&](int tile, int slot) { // 2 K pieces (this wave's share of the tile) + 4 dS pieces (its own two tiles)
        const int tl = tile < nt ? tile : nt - 1;
        const int kfirst = tl * ABK, klast = (int)a.Skv - 1 - kfirst; // (rows of this tile that exist: 0 .. klast)
        const char *kg = Kg + (int64_t)kfirst * a.lk.sr;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row0 = (wid * 2 + i) * 4, row = row0 + krow;
            const int chunk = kpos ^ (((row & 3) << 2) | ((row >> 2) & 3));
            // a key row beyond Skv (ragged last tile) is fetched from the tensor's LAST row instead: its dS column is exactly zero (it lies above every
            // query's diagonal), so any finite K row serves - and that one is always there
            const int srow = row < klast ? row : klast;
            if (D == AD || chunk < D / 8) // head size 64: see f_stage
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(kg + (int64_t)srow * a.lk.sr + chunk * 16),
                                                 (__attribute__((address_space(3))) void *)(smem + slot * FTILE + row0 * AROW), 16, 0, 0);
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            int kwb = 2 * tl + hf;
            kwb = kwb < kwb_last ? kwb : kwb_last;
            // key blocks above this wave's slice are never consumed and have no tile (only the causal half is kept): fetch the slice's own last
            // tile again instead - it came through a few steps ago, the bytes come from cache, not from HBM (this kernel is bound by its dS stream)
            kwb = kwb <= slw ? kwb : (slw < kwb_last ? slw : kwb_last);
            const char *tg = dsg + (int64_t)kwb * 8 * DS_TILE;
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(tg + ds_src[pc]),
                                                 (__attribute__((address_space(3))) void *)(slab + slot * DQ_SLAB + hf * DS_TILE + pc * 1024), 16, 0, 0);
        }
    };
    stage(0, 0);
    stage(1, 1);
    for (int t = 0; t < nt; ++t) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // step t has landed (this wave's pieces) ...
        __builtin_amdgcn_s_barrier();                      // ... and the K tile's other pieces; slot (t + 2) % 3 is free again
        asm volatile("" ::: "memory");
        stage(t + 2, (t + 2) % DQ_RING);
        const char *kt = smem + (t % DQ_RING) * FTILE;
        const unsigned dsb = ds_rd + (unsigned)((t % DQ_RING) * DQ_SLAB);
        if (active && 2 * t <= sl && !KF_MUT(3, t == 0 && qblk == nxb - 1)) { // wave-uniform: key block 2 t lies at or below this slice's diagonal
            dq_step<BF, 0, 0, DB>(kt, vo, dsb, dq);
            dq_step<BF, 0, 1, DB>(kt, vo, dsb, dq);
            if (2 * t + 1 <= sl) {
                dq_step<BF, 1, 0, DB>(kt, vo, dsb, dq);
                dq_step<BF, 1, 1, DB>(kt, vo, dsb, dq);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (active) a_store_rows<BF, DB>(smem + wid * 32 * OPAD, a.dq + a_head(a.ldq, bh, a.H) + qw * a.ldq.sr, dq, a.scale, a.ldq.sr,
                                     (int)(a.Sq - qw < 32 ? a.Sq - qw : 32));
    if (a.persist) __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------
// backward: dK, dV ("key on the lane"); both kernels sweep 32-query slices from the block's diagonal down. Per slice:
// S = Q K^T, dP = dO V^T (A = Q / dO rows from LDS, B = K / V fragments), dV^T += dO^T P, dK^T += Q^T dS (A via
// transposed reads).
// ------------------------------------------------------------------------------------------
constexpr int BQS = 32;  // queries per slice

// ==========================================================================================
// backward: dK, dV, v4 - 4 waves x 32 keys, ONE wave per SIMD and the 512-register budget that goes with it.
// A two-waves-per-SIMD form (round 1's first dK/dV kernel, removed) cannot keep enough LDS reads in flight (256 registers,
// 128 of them accumulators): its waves sat in s_waitcnt two thirds of the time. Here a wave walks a slice as eight quarter-phases
// of 4 MFMAs (S, S, dP, dP, dV, dV, dK, dK), every quarter-phase's LDS fragments requested two quarter-phases ahead
// (inline-asm reads, counted lgkmcnt waits that name their destinations). The arithmetic hides under MFMAs of the
// SAME slice: p = exp2(c S') needs only S, so it runs under the dP chain; dV needs only p, so dS = p dP' runs under
// the dV MFMAs. K and V fragments stay in registers (B operands of S and dP).
// Row constants as initial accumulators: the pre-pass stores -lse log2(e) and -delta, a slice's 2 x 32 of them come
// into LDS with its tiles and start the S and dP chains (head size 128: as the C operand of a chain's first MFMA; head size 64:
// read into the accumulator), so p = exp2(S'') and dS = p dP'.
// Head size 128 runs the S and dP chains as inline-asm MFMAs on VGPR accumulators (k4_mfma_first / k4_mfma_acc below: what that
// buys, and the hazards it makes ours); its LDS reads go one or two behind every MFMA. Head size 64 keeps the builtin form.
// LDS image of a 32-row tile: 8-row x 32-column subtiles of 512 B with the chunk XOR inside a subtile (guide T10,
// image (a)) - two base VGPRs serve the 8 row reads of a tile and two the 16 transposed reads, the rest are
// immediates. Q / dO slices stream through a ring of four slice PAIRS by LDS-DMA, one DMA operation per quarter-phase;
// one barrier per pair, placed two quarter-phases before the pair ends so the next pair's first fragments are already
// in flight at the loop edge.
// ==========================================================================================
constexpr int K4B = 128;                        // keys per block
constexpr int K4SL = 2 * BQS * AROW + 256;      // slice buffer: Q tile | dO tile | nlse[32] | ndelta[32]
constexpr int K4PAIR = 2 * K4SL;
constexpr int K4LDS = 4 * K4PAIR;               // ring of four slice pairs, 130 KiB
constexpr bool K4_SPREAD = true; // DMA operations one per MFMA behind the barrier (false: all ten at once)
static_assert(K4LDS >= 4 * 32 * OPAD, "epilogue slabs must fit");

template <int OFF>
__device__ __forceinline__ void k4_rows4(unsigned e, unsigned o, s16x8 (&f)[4]) { // four consecutive k-steps of one tile: even off e, odd off o
    asm volatile("ds_read_b128 %0, %4 offset:%c6\n\tds_read_b128 %1, %5 offset:%c6\n\t"
                 "ds_read_b128 %2, %4 offset:%c7\n\tds_read_b128 %3, %5 offset:%c7"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
                 : "v"(e), "v"(o), "n"(OFF), "n"(OFF + 512)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void k4_rowc(unsigned lr, f32x4 (&c)[4]) { // 16 row constants: rows 8 g + 4 hl + {0..3}
    asm volatile("ds_read_b128 %0, %4 offset:%c5\n\tds_read_b128 %1, %4 offset:%c6\n\t"
                 "ds_read_b128 %2, %4 offset:%c7\n\tds_read_b128 %3, %4 offset:%c8"
                 : "=&a"(c[0]), "=&a"(c[1]), "=&a"(c[2]), "=&a"(c[3]) // straight into accumulator registers
                 : "v"(lr), "n"(OFF), "n"(OFF + 32), "n"(OFF + 64), "n"(OFF + 96)
                 : "memory");
}
struct K4Tr { s16x4 lo[4], hi[4]; };
template <int OFF>
__device__ __forceinline__ void k4_tr4(unsigned t0, unsigned t1, K4Tr &t) { // column blocks d = 0..3 of one 16-row k-step
    asm volatile("ds_read_b64_tr_b16 %0, %8 offset:%c10\n\tds_read_b64_tr_b16 %1, %9 offset:%c10\n\t"
                 "ds_read_b64_tr_b16 %2, %8 offset:%c11\n\tds_read_b64_tr_b16 %3, %9 offset:%c11\n\t"
                 "ds_read_b64_tr_b16 %4, %8 offset:%c12\n\tds_read_b64_tr_b16 %5, %9 offset:%c12\n\t"
                 "ds_read_b64_tr_b16 %6, %8 offset:%c13\n\tds_read_b64_tr_b16 %7, %9 offset:%c13"
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1]), "=&v"(t.lo[2]), "=&v"(t.hi[2]), "=&v"(t.lo[3]), "=&v"(t.hi[3])
                 : "v"(t0), "v"(t1), "n"(OFF), "n"(OFF + 512), "n"(OFF + 1024), "n"(OFF + 1536)
                 : "memory");
}
// counted waits: N = LDS operations issued after the ones being waited for; the destinations are named so that no
// consumer can be scheduled above the wait
template <int N>
__device__ __forceinline__ void k4_wait4(s16x8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait4c(s16x8 (&f)[4], f32x4 (&c)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c8)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+a"(c[0]), "+a"(c[1]), "+a"(c[2]), "+a"(c[3])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait_tr(K4Tr &a) {
    asm volatile("s_waitcnt lgkmcnt(%c8)"
                 : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]), "+v"(a.lo[2]), "+v"(a.hi[2]), "+v"(a.lo[3]), "+v"(a.hi[3])
                 : "n"(N)
                 : "memory");
}
template <bool BF>
__device__ __forceinline__ typename AFrag<BF>::type k4_frag(const K4Tr &t, int d) {
    s16x8 r;
    r[0] = t.lo[d][0]; r[1] = t.lo[d][1]; r[2] = t.lo[d][2]; r[3] = t.lo[d][3];
    r[4] = t.hi[d][0]; r[5] = t.hi[d][1]; r[6] = t.hi[d][2]; r[7] = t.hi[d][3];
    return __builtin_bit_cast(typename AFrag<BF>::type, r);
}
__device__ __forceinline__ f32x16 k4_acc(const f32x4 (&c)[4]) {
    f32x16 r;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[4 * g + j] = c[g][j];
    return r;
}

// ---- head size 128: the S and dP chains as inline-asm MFMAs whose accumulator is a VGPR tuple ----------------------------------
// hipcc gives every MFMA *builtin* of a kernel that uses accumulator registers an AGPR destination, and the VALU cannot read an
// AGPR: each exp2 / multiply of a score then costs a v_accvgpr_read on top, and the row constants reach the accumulators through
// v_accvgpr_mov (136 of the 274 VALU instructions of a slice pair; with one wave per SIMD a wave's VALU instructions do not run
// under its own MFMAs - tools/scratch/mfma_valu_overlap.hip - so they are time). Written as asm the chains keep S and dP in VGPRs
// (read in place), start from the row constants as the C operand of their first MFMA (no copies), and take the K / V operands
// from AGPRs, which is where the 64 registers of those fragments now live (only MFMAs read them).
// Hazards the compiler cannot see inside asm: an 8-pass MFMA's VGPR result may be read by the VALU 11 wait states later
// (GFX940 XDL write -> VALU read); every first read below sits behind >= 8 LDS instructions, a wait and an MFMA, plus the s_nop
// of the pin that follows; the first MFMA of a chain carries an s_nop 1 in front for a C operand the compiler might have just moved.
// And the other direction: to the compiler an asm statement is done with its inputs when it is issued, so it would hand an A or C
// operand's registers to the very next VALU result (it did: v_exp_f32 into the first register of the fragment an MFMA two
// instructions earlier was still reading - NaNs in one accumulator element of the masked loop only). The VGPR operands are
// therefore in-out operands of the MFMA statements and stay live until k4_keep() names them, a quarter-phase later.
template <int OFF>
__device__ __forceinline__ void k4_rows4a(unsigned e, unsigned o, s16x8 (&f)[4]) { // k4_rows4 into accumulator registers
    asm volatile("ds_read_b128 %0, %4 offset:%c6\n\tds_read_b128 %1, %5 offset:%c6\n\t"
                 "ds_read_b128 %2, %4 offset:%c7\n\tds_read_b128 %3, %5 offset:%c7"
                 : "=&a"(f[0]), "=&a"(f[1]), "=&a"(f[2]), "=&a"(f[3])
                 : "v"(e), "v"(o), "n"(OFF), "n"(OFF + 512)
                 : "memory");
}
template <int OFF>
__device__ __forceinline__ void k4_rowcv(unsigned lr, f32x4 (&c)[4]) { // k4_rowc into VGPR quads
    asm volatile("ds_read_b128 %0, %4 offset:%c5\n\tds_read_b128 %1, %4 offset:%c6\n\t"
                 "ds_read_b128 %2, %4 offset:%c7\n\tds_read_b128 %3, %4 offset:%c8"
                 : "=&v"(c[0]), "=&v"(c[1]), "=&v"(c[2]), "=&v"(c[3])
                 : "v"(lr), "n"(OFF), "n"(OFF + 32), "n"(OFF + 64), "n"(OFF + 96)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait4a_cv(s16x8 (&f)[4], f32x4 (&c)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c8)"
                 : "+a"(f[0]), "+a"(f[1]), "+a"(f[2]), "+a"(f[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "n"(N)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait4a(s16x8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c4)" : "+a"(f[0]), "+a"(f[1]), "+a"(f[2]), "+a"(f[3]) : "n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait4_cv(s16x8 (&f)[4], f32x4 (&c)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c8)"
                 : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]), "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3])
                 : "n"(N)
                 : "memory");
}
// The slice's LDS reads one or two per MFMA gap instead of four to twelve at the head of a quarter-phase (a lone wave hides about
// five single-issue instructions behind an MFMA; a burst of reads in one gap is paid in full: MI355X_MICROARCH.md, "one wave per
// SIMD ... HIDDEN per MFMA gap"). Pieces of k4_rows4 / k4_rowc / k4_tr4:
template <int OFF>
__device__ __forceinline__ void k4_row1(unsigned a, s16x8 &f) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(f) : "v"(a), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void k4_row1a(unsigned a, s16x8 &f) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&a"(f) : "v"(a), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void k4_rowc1(unsigned lr, f32x4 &c) { asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=&v"(c) : "v"(lr), "n"(OFF) : "memory"); }
template <int OFF>
__device__ __forceinline__ void k4_trp(unsigned t0, unsigned t1, s16x4 &lo, s16x4 &hi) { // one column block of a 16-row k-step
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c4\n\tds_read_b64_tr_b16 %1, %3 offset:%c4" : "=&v"(lo), "=&v"(hi) : "v"(t0), "v"(t1), "n"(OFF) : "memory");
}
// acc = A B + C (first MFMA of a chain; A fragment in AGPRs (AA) or VGPRs), acc += A B (the rest)
template <bool BF, bool AA>
__device__ __forceinline__ void k4_mfma_first(f32x16 &acc, s16x8 &a, const typename AFrag<BF>::type &b, f32x16 &c) {
    if constexpr (BF) {
        if constexpr (AA) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %2" : "=&v"(acc), "+a"(a), "+v"(c) : "a"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %3, %2" : "=&v"(acc), "+v"(a), "+v"(c) : "a"(b));
    } else {
        if constexpr (AA) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %3, %2" : "=&v"(acc), "+a"(a), "+v"(c) : "a"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %3, %2" : "=&v"(acc), "+v"(a), "+v"(c) : "a"(b));
    }
}
template <bool BF, bool AA>
__device__ __forceinline__ void k4_mfma_acc(f32x16 &acc, s16x8 &a, const typename AFrag<BF>::type &b) {
    if constexpr (BF) {
        if constexpr (AA) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc), "+a"(a) : "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc), "+v"(a) : "a"(b));
    } else {
        if constexpr (AA) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc), "+a"(a) : "a"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc), "+v"(a) : "a"(b));
    }
}
// the same with a second VGPR tuple as an in-out operand that the instruction does not touch: the arithmetic on `tag` (exp2 of S
// beside the dP chain) then stays between two MFMAs of the chain without separate ordering statements (hipcc pads every inline
// asm whose registers a neighbouring instruction touches with a wait state: three s_nop per MFMA gap with the pins, one without)
template <bool BF>
__device__ __forceinline__ void k4_mfma_first_t(f32x16 &acc, s16x8 &a, const typename AFrag<BF>::type &b, f32x16 &c, f32x16 &tag) {
    if constexpr (BF) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %4, %2\n\ts_nop 3" : "=&v"(acc), "+v"(a), "+v"(c), "+v"(tag) : "a"(b));
    else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %4, %2\n\ts_nop 3" : "=&v"(acc), "+v"(a), "+v"(c), "+v"(tag) : "a"(b));
}
template <bool BF>
__device__ __forceinline__ void k4_mfma_acc_t(f32x16 &acc, s16x8 &a, const typename AFrag<BF>::type &b, f32x16 &tag) {
    if constexpr (BF) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %3, %0" : "+v"(acc), "+v"(a), "+v"(tag) : "a"(b));
    else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %3, %0" : "+v"(acc), "+v"(a), "+v"(tag) : "a"(b));
}
// the registers of these operands may be reused only after this point
__device__ __forceinline__ void k4_keep(s16x8 (&f)[4]) { asm volatile("" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3])); }
__device__ __forceinline__ void k4_keep(f32x16 &c) { asm volatile("" : "+v"(c)); }

// two column blocks of one 16-row k-step (head size 64)
template <int OFF>
__device__ __forceinline__ void k4_tr2(unsigned t0, unsigned t1, K4Tr &t) {
    asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%c6\n\tds_read_b64_tr_b16 %1, %5 offset:%c6\n\t"
                 "ds_read_b64_tr_b16 %2, %4 offset:%c7\n\tds_read_b64_tr_b16 %3, %5 offset:%c7"
                 : "=&v"(t.lo[0]), "=&v"(t.hi[0]), "=&v"(t.lo[1]), "=&v"(t.hi[1])
                 : "v"(t0), "v"(t1), "n"(OFF), "n"(OFF + 512)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void k4_wait_tr2(K4Tr &a) {
    asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(a.lo[0]), "+v"(a.hi[0]), "+v"(a.lo[1]), "+v"(a.hi[1]) : "n"(N) : "memory");
}

// one v_mul_f32, opaque to the vectoriser: beside MFMAs a v_pk_mul_f32 costs more than the two multiplies it replaces
// (MI355X_MICROARCH.md, "packed f32 VALU ... an anti-lever beside MFMAs"), and -O3 packs adjacent tuple elements on its own
// (the odd element of a pair as a x b + 0: a v_fma_f32 beside a v_mul_f32 is not a pair the vectoriser packs, and neither is inline
// asm, around which hipcc would put wait states)
__device__ __forceinline__ float k4_mul(float a, float b) { return a * b; }
__device__ __forceinline__ float k4_mul_odd(float a, float b) { return __builtin_fmaf(a, b, 0.f); }
#ifdef KF_ABL_DKV_NOVALU // ablation (scratch builds only): the slice without its exp2 and its dS multiplies
#define KF_ABL_EXP(x) (x)
#define KF_ABL_MUL(a, b) (b)
#define KF_ABL_MUL_ODD(a, b) (b)
#else
#define KF_ABL_EXP(x) __builtin_amdgcn_exp2f(x)
#define KF_ABL_MUL(a, b) k4_mul(a, b)
#define KF_ABL_MUL_ODD(a, b) k4_mul_odd(a, b)
#endif
template <bool BF, bool DS, int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_v4_kernel(const AttnArgs a) {
    using frag_t = typename AFrag<BF>::type;
    constexpr int KS = D / 16, DB = D / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), xl = lane & 31, hl = lane >> 5;
    int xb0;
    int64_t bh;
    // a.persist: a workgroup takes key block x and its causal mirror nkb - 1 - x (equal work per workgroup, as in the forward)
    const int nkb = (int)(a.Skv / K4B), nwx = a.persist ? nkb / (2 * a.persist) : nkb;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    bh += a.bh0;
    const char *Qg = a.q + a_head(a.lq, bh, a.H);
    const char *dOg = a.d_o + a_head(a.ldo, bh, a.H);
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nkb - 1 - xp : xp;
    const int64_t k0 = (int64_t)xb * K4B, kw = k0 + wid * 32, n = kw + xl;
#ifdef KF_ATTN_TIMELINE
    const unsigned long long tl_pass0 = __builtin_amdgcn_s_memtime();
#endif

    frag_t kf[8], vf[8]; // this wave's 32 keys: B operands of S = Q K^T and dP = dO V^T
    const float c = a.scale * kLog2e;
    {
        const char *Kg = a.k + a_head(a.lk, bh, a.H) + n * a.lk.sr;
        const char *Vg = a.v + a_head(a.lv, bh, a.H) + n * a.lv.sr;
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
            kf[kk] = *(const frag_t *)(Kg + (kk * 16 + 8 * hl) * 2);
            vf[kk] = *(const frag_t *)(Vg + (kk * 16 + 8 * hl) * 2);
        }
    }
    f32x16 dk[4], dv[4]; // head size 64 uses column blocks 0 and 1
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }

    // ---- per-lane LDS read addresses inside a slice buffer (tile image (a): off(row, ch) =
    //      2048 (row >> 3) + 512 (ch >> 2) + 64 (row & 7) + 16 ((ch & 3) ^ ((row >> 2) & 3)))
    const unsigned smem_u = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    const int rsw = (xl >> 2) & 3;
    const unsigned rb_e = smem_u + 2048 * (xl >> 3) + 64 * (xl & 7) + 16 * ((0 + hl) ^ rsw); // row xl, chunk 2 kk + hl, kk even
    const unsigned rb_o = smem_u + 2048 * (xl >> 3) + 64 * (xl & 7) + 16 * ((2 + hl) ^ rsw); // kk odd
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3, h = g >> 1;
    const int tslot = 2 * (g & 1) + (pp >> 1);
    const unsigned tb_0 = smem_u + 64 * (4 * h + qq) + 16 * (tslot ^ h) + 8 * (pp & 1);              // rows 16 ks + 4 h + qq
    const unsigned tb_1 = smem_u + 2048 + 64 * (4 * h + qq) + 16 * (tslot ^ (h + 2)) + 8 * (pp & 1); // ... + 8
    const unsigned lr = smem_u + 2 * BQS * AROW + 16 * hl;                                           // row constants 8 g4 + 4 hl

    // ---- LDS-DMA: a 32-row tile is 8 wave-instructions of 1 KiB (8 rows x 128 B each); wave w moves rows 8 w .. 8 w + 7
    // of the Q tile and of the dO tile, and every wave fetches the 64 row constants (identical bytes: uniform counts).
    int64_t soffq[2], soffd[2]; // per-lane source offsets inside a slice: row * (row stride of Q | dO) + chunk
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int sub = 2 * i + (lane >> 5), r = (lane >> 2) & 7, slot = lane & 3, row = 8 * wid + r;
        const int ch = (4 * sub + (slot ^ ((row >> 2) & 3))) * 16;
        soffq[i] = row * a.lq.sr + ch;
        soffd[i] = row * a.ldo.sr + ch;
    }
    const float *rcg = lane < BQS ? a.nlse + bh * a.Sqc + lane : a.ndelta + bh * a.Sqc + lane - BQS;
    const int ns = (int)(a.Sq / BQS), np = ns / 2;
    // dS tiles of this wave's 32 keys: tile (qb, kwb, sl) of the workspace, lane (key xl, half hl) writes operand s at
    // s * 1024 + xl * 32 + hl * 16 (see DS_TILE)
    unsigned ds_lane = (unsigned)(xl * 32 + hl * 16);
    // this wave's 32-key block kwb = kw / 32 of the pair: tile (kwb, sl) = ds_tile_index = row of query block sl / 8 + kwb * 8 + (sl & 7)
    const char *ds_base = DS ? a.ds + ((bh - a.bh0) * a.ds_pair + (kw >> 5) * 8) * DS_TILE : nullptr;
    const int ds_nkb = (int)a.ds_nkb, ds_tri = a.ds_tri;
    // a pair is 10 DMA operations per wave (ids 0..9: slice id / 5; Q rows i, dO rows i for i = 0, 1, then the row
    // constants); they are issued ONE per quarter-phase (an LDS-DMA instruction holds the wave's issue for 60-180 cycles,
    // which a lone wave per SIMD can only hide under MFMAs already queued)
    auto stage_piece = [&](auto id_c, int pr_, int slot4) __attribute__((always_inline)) {
        constexpr int ID = decltype(id_c)::value, SL = ID / 5, K = ID % 5;
        if constexpr (D == 64 && (K == 2 || K == 3)) return; // head size 64: the rows end after column blocks 0 and 1 (pieces i = 0)
        const int prc = pr_ < np ? pr_ : np - 1; // past the end: re-fetch the last pair (keeps the counts uniform; never consumed)
        const int64_t qs_ = ((int64_t)prc * 2 + SL) * BQS;
        char *buf = smem + slot4 * K4PAIR + SL * K4SL;
        if constexpr (K == 4) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rcg + qs_),
                                             (__attribute__((address_space(3))) void *)(buf + 2 * BQS * AROW), 4, 0, 0);
        } else {
            constexpr int i = K >> 1;
            const char *src = (K & 1) ? dOg + qs_ * a.ldo.sr + soffd[i] : Qg + qs_ * a.lq.sr + soffq[i];
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(buf + (K & 1) * BQS * AROW + 1024 * (2 * wid + i)), 16, 0, 0);
        }
    };
    auto stage_pair = [&](int pr_, int slot4) __attribute__((always_inline)) {
        stage_piece(std::integral_constant<int, 0>{}, pr_, slot4); stage_piece(std::integral_constant<int, 1>{}, pr_, slot4);
        stage_piece(std::integral_constant<int, 2>{}, pr_, slot4); stage_piece(std::integral_constant<int, 3>{}, pr_, slot4);
        stage_piece(std::integral_constant<int, 4>{}, pr_, slot4); stage_piece(std::integral_constant<int, 5>{}, pr_, slot4);
        stage_piece(std::integral_constant<int, 6>{}, pr_, slot4); stage_piece(std::integral_constant<int, 7>{}, pr_, slot4);
        stage_piece(std::integral_constant<int, 8>{}, pr_, slot4); stage_piece(std::integral_constant<int, 9>{}, pr_, slot4);
    };

    const int p0 = (int)(k0 / (2 * BQS)); // first pair holding a query >= the block's first key
    stage_pair(p0, 0);
    stage_pair(p0 + 1, 1);
    stage_pair(p0 + 2, 2);
    { // (exact f32 scores since round 4: K stays as it is, the row constant is -lse / scale and every score is multiplied by c = scale log2(e) in
      // front of its exp2. Rounds 2-3 scaled K by c in 16 bits here instead - 32 VALU instructions fewer per slice pair, and a score error of
      // eps scale sum |q k| that left the parity bounds once the logits grew: profiles/r04_attn_large_logits.txt)
        // only MFMAs read them from here on: accumulator registers
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) asm volatile("" : "+a"(kf[kk]), "+a"(vf[kk]));
    }
    // pairs p0 and p0 + 1 have landed (this wave's part: everything but the DMA operations of the third pair; the counted wait of
    // the loop assumes a pair's predecessor-but-one complete, which the first iteration gets from here) ...
    if constexpr (D == 128) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();                      // ... and everyone else's
    asm volatile("" ::: "memory");

    // One slice = eight quarter-phases of 4 MFMAs, each with its own LDS read group requested two quarter-phases ahead:
    //   q0 S k0..3 [Q 0..3]   q1 S k4..7 [Q 4..7]   q2 dP k0..3 [dO 0..3]   q3 dP k4..7 [dO 4..7]
    //   q4 dV k-step 0 [dO^T]  q5 dV k-step 1 [dO^T]  q6 dK k-step 0 [Q^T]  q7 dK k-step 1 [Q^T]
    // in issue order  q5: dO^T k1 + next slice's S constants (12) | q6: next Q 0..3 (4) | q7: next Q 4..7 + next dP constants
    // (8) | q0: dO 0..3 (4) | q1: dO 4..7 (4) | q2: dO^T k0 (8) | q3: dO^T... (8) | q4: Q^T k0 (8); a wait leaves the two
    // younger groups in flight (lgkmcnt saturates at 15).
    // Every VALU instruction of the slice rides behind an MFMA of the same wave, in an order pinned by empty volatile asm
    // statements (MFMA and VALU intrinsics are pure values, sched_barrier does not order them; left alone the scheduler
    // lumps the arithmetic between the MFMA groups and a lone wave then leaves the matrix pipe idle meanwhile):
    //   q1  the dP accumulator takes its row constants        q2, q3  p = exp2(c S'), two elements per MFMA; pack p (k0)
    //   q4  dS = p dP' (k0 elements); pack p (k1)               q5  dS (k1 elements); pack dS (k0)
    //   q6  pack dS (k1)                                        q7  the next slice's S accumulator takes its row constants
#ifdef KF_ATTN_TIMELINE
    unsigned long long tlk[10], tlk_acc[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlk_prev = __builtin_amdgcn_s_memtime();
#endif
    s16x8 g0[4], g1[4];
    f32x4 cs[4], cp[4];
    if constexpr (D == 128) { // the first slice's first groups: constants into VGPR quads, Q rows into accumulator registers
        k4_rowcv<0>(lr, cs);
        k4_rows4a<0>(rb_e, rb_o, g0);
        k4_rows4a<1024>(rb_e, rb_o, g1);
        k4_rowcv<128>(lr, cp);
    } else { // head size 64: constants, Q rows (accumulator registers), dP constants, dO rows (carried in g1)
        k4_rowcv<0>(lr, cs);
        k4_rows4a<0>(rb_e, rb_o, g0);
        k4_rowcv<128>(lr, cp);
        k4_rows4<BQS * AROW>(rb_e, rb_o, g1);
    }

#define K4_MFMA4(ACC, FR, BOP, K0)                                                                                   \
    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) ACC = a_mfma<BF>(__builtin_bit_cast(frag_t, FR[kk]), BOP[K0 + kk], ACC);
#define K4_CVT2(DST, J, A, B)                                              \
    if constexpr (BF) { DST[J] = (__bf16)(A); DST[(J) + 1] = (__bf16)(B); } \
    else { DST[J] = (_Float16)(A); DST[(J) + 1] = (_Float16)(B); }
// the P pack: f16 carries P as P 2^14 (kPShiftF16: <= 16384, cannot overflow; the format's subnormal range then starts at 3.7e-9 instead of
// 6.1e-5 - at large logits whole key columns of P lay below that and dV lost them); dV is scaled back when it is stored
#define K4_CVTP(DST, J, A, B)                                              \
    if constexpr (BF) { DST[J] = (__bf16)(A); DST[(J) + 1] = (__bf16)(B); } \
    else { DST[J] = (_Float16)((A) * kPShiftF16); DST[(J) + 1] = (_Float16)((B) * kPShiftF16); }

    // SOFF: this slice's offset inside its pair buffer (bases e, o, t0, t1, l); the next slice's first groups are read off
    // (en, on, ln) + NOFF; LAST: the slice that ends a pair (barrier + DMA of the pair after next before its q6)
    auto slice_body = [&, ds_lane, ds_base, ds_nkb, ds_tri](auto mask_c, auto soff_c, auto noff_c, auto last_c, unsigned e, unsigned o, unsigned t0,
                          unsigned t1, unsigned l, unsigned en, unsigned on, unsigned ln, int64_t qs, int pr, int it) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_c)::value, LAST = decltype(last_c)::value;
        constexpr int SOFF = decltype(soff_c)::value, NOFF = decltype(noff_c)::value, DO = SOFF + BQS * AROW;
        // LDS reads, one or two behind every MFMA, each group requested during the quarter-phase TWO before the one that consumes it:
        //   q0: g2 (dO rows k 0..3)   q1: g3 (dO rows k 4..7)   q2: t4 (dO^T k-step 0)   q3: t5 (dO^T k-step 1)   q4: t6 (Q^T 0)   q5: t7 (Q^T 1)
        //   q6: the next slice's S constants + its Q rows k 0..3 (after the pair barrier)   q7: its Q rows k 4..7 + its dP constants
        // so a quarter-phase opens with one counted wait that leaves exactly the previous quarter-phase's group in flight
        // (lgkmcnt 8, 4, 4, 8, 8, 8, 8, 8 for q0 .. q7).
        f32x16 sv, dpv; // VGPR tuples: written by the asm MFMA chains; p = exp2(S'') replaces S in place and dS = p dP' replaces dP
        const int nd = (int)(n - qs) - 4 * hl; // key - query of accumulator element e is nd - a_row(e, 0): masked when positive
        s16x8 g2[4], g3[4];
        K4Tr t4, t5, t6, t7;
        frag_t pf[2], df[2];
        TLK_STAMP(0)
        // q0: S k-steps 0..3
        k4_wait4a_cv<8>(g0, cs);
        f32x16 c_s = k4_acc(cs), c_p;
        k4_mfma_first<BF, true>(sv, g0[0], kf[0], c_s); // C = - lse log2(e): the chain starts from the row constants
        k4_row1<DO>(e, g2[0]);
        k4_mfma_acc<BF, true>(sv, g0[1], kf[1]);
        k4_row1<DO>(o, g2[1]);
        k4_mfma_acc<BF, true>(sv, g0[2], kf[2]);
        k4_row1<DO + 512>(e, g2[2]);
        k4_mfma_acc<BF, true>(sv, g0[3], kf[3]);
        k4_row1<DO + 512>(o, g2[3]);
        TLK_STAMP(1)
        // q1: S k-steps 4..7
        k4_wait4a<4>(g1);
        k4_mfma_acc<BF, true>(sv, g1[0], kf[4]);
        k4_row1<DO + 1024>(e, g3[0]);
        k4_mfma_acc<BF, true>(sv, g1[1], kf[5]);
        k4_row1<DO + 1024>(o, g3[1]);
        k4_mfma_acc<BF, true>(sv, g1[2], kf[6]);
        k4_row1<DO + 1536>(e, g3[2]);
        k4_mfma_acc<BF, true>(sv, g1[3], kf[7]);
        k4_row1<DO + 1536>(o, g3[3]);
        k4_keep(c_s);
        TLK_STAMP(2)
        // q2: dP k-steps 0..3; p = exp2(S'') for elements 0..7
        k4_wait4_cv<4>(g2, cp);
#define K4_EXP2(E)                                                                      \
    _Pragma("unroll") for (int e_ = (E); e_ < (E) + 2; ++e_) {                          \
        float pv = KF_ABL_EXP(sv[e_] * c);                                              \
        if (MASK && nd > a_row(e_, 0)) pv = 0.f;                                        \
        if (KF_MUT(2, xb == 0 && qs == 4 * BQS)) pv = 0.f;                              \
        sv[e_] = pv;                                                                    \
    }
#define K4_Q2(KK)                                                                                       \
    if constexpr ((KK) == 0) {                                                                          \
        c_p = k4_acc(cp);                                                                               \
        k4_mfma_first_t<BF>(dpv, g2[0], vf[0], c_p, sv); /* C = - delta; S is read from here on: hazard note */ \
    } else {                                                                                            \
        k4_mfma_acc_t<BF>(dpv, g2[KK], vf[KK], sv);                                                     \
    }                                                                                                   \
    K4_EXP2(2 * (KK))                                                                                   \
    k4_trp<DO + 512 * (KK)>(t0, t1, t4.lo[KK], t4.hi[KK]);
        K4_Q2(0) K4_Q2(1) K4_Q2(2) K4_Q2(3)
#undef K4_Q2
        TLK_STAMP(3)
        // q3: dP k-steps 4..7; elements 8..15; pack p of k-step 0
        k4_wait4<8>(g3);
#define K4_Q3(KK)                                                                                       \
    k4_mfma_acc_t<BF>(dpv, g3[KK], vf[4 + (KK)], sv);                                                   \
    K4_EXP2(8 + 2 * (KK))                                                                               \
    K4_CVTP(pf[0], 2 * (KK), sv[2 * (KK)], sv[2 * (KK) + 1])                                            \
    k4_trp<DO + 4096 + 512 * (KK)>(t0, t1, t5.lo[KK], t5.hi[KK]);
        K4_Q3(0) K4_Q3(1) K4_Q3(2) K4_Q3(3)
#undef K4_Q3
#undef K4_EXP2
        k4_keep(c_p);
        k4_keep(g2);
        TLK_STAMP(4)
        // q4: dV k-step 0; dS = p dP' for elements 0..7; pack p of k-step 1 (s_nop: the dP chain's result is read three instructions on;
        // tools/kernel_hazards.py counts the wait states of every such pair in the compiled kernel)
        k4_wait_tr<8>(t4);
        asm volatile("s_nop 2" : "+v"(dpv));
#define K4_Q4(DD)                                                                                                       \
    dv[DD] = a_mfma<BF>(k4_frag<BF>(t4, DD), pf[0], dv[DD]);                                                            \
    asm volatile("" : "+a"(dv[DD]), "+v"(dpv), "+v"(sv));                                                               \
    dpv[2 * (DD)] = KF_ABL_MUL(sv[2 * (DD)], dpv[2 * (DD)]); dpv[2 * (DD) + 1] = KF_ABL_MUL_ODD(sv[2 * (DD) + 1], dpv[2 * (DD) + 1]); \
    K4_CVTP(pf[1], 2 * (DD), sv[8 + 2 * (DD)], sv[9 + 2 * (DD)])                                                        \
    asm volatile("" : "+a"(dv[((DD) + 1) & 3]), "+v"(dpv), "+v"(pf[1]));                                                \
    k4_trp<SOFF + 512 * (DD)>(t0, t1, t6.lo[DD], t6.hi[DD]);
        K4_Q4(0) K4_Q4(1) K4_Q4(2) K4_Q4(3)
#undef K4_Q4
        k4_keep(g3);
        TLK_STAMP(5)
        // q5: dV k-step 1; dS for elements 8..15; pack dS of k-step 0
        k4_wait_tr<8>(t5);
#define K4_Q5(DD)                                                                                                       \
    dv[DD] = a_mfma<BF>(k4_frag<BF>(t5, DD), pf[1], dv[DD]);                                                            \
    asm volatile("" : "+a"(dv[DD]), "+v"(dpv));                                                                         \
    dpv[8 + 2 * (DD)] = KF_ABL_MUL(sv[8 + 2 * (DD)], dpv[8 + 2 * (DD)]); dpv[9 + 2 * (DD)] = KF_ABL_MUL_ODD(sv[9 + 2 * (DD)], dpv[9 + 2 * (DD)]); \
    K4_CVT2(df[0], 2 * (DD), dpv[2 * (DD)], dpv[2 * (DD) + 1])                                                          \
    asm volatile("" : "+a"(dv[((DD) + 1) & 3]), "+v"(dpv), "+v"(df[0]));                                                \
    k4_trp<SOFF + 4096 + 512 * (DD)>(t0, t1, t7.lo[DD], t7.hi[DD]);
        K4_Q5(0) K4_Q5(1) K4_Q5(2) K4_Q5(3)
#undef K4_Q5
        TLK_STAMP(6)
        // q6: dK k-step 0; pack dS of k-step 1
        if constexpr (LAST) {
            // the next pair has landed - this wave's part, then everyone's - and every wave is past its reads of the
            // previous pair, whose buffer takes pair pr + 3. vmcnt(0), not a counted wait: register spills are VMEM
            // operations too and would be counted among "the youngest"; the pair after next was issued a pair ago.
            // With the dS stores in the stream (two per slice, one in q6 and one in q7) the wait is COUNTED: the pair being waited
            // for (pr + 1) was issued in pair pr - 2's last slice, whose last operation is its last DMA piece; younger than that are
            // pair pr - 1's 2 + (10 + 2) and this pair's first slice's 2 = 16 operations (a store that has just been issued takes
            // ~1 us to retire: vmcnt(0) here would stall every pair on it). The kernel must stay spill-free for this to hold
            // (tools/kernel_resources.py). Every read of the NEXT pair's buffer (the constants and Q rows below) comes after this barrier.
            if constexpr (DS) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (!K4_SPREAD) stage_pair(pr + 3, (it + 3) & 3);
        }
        TLK_STAMP(8) // (LAST slices: after the counted vmcnt wait and the barrier)
        // the slice's dS, already packed as two MFMA operands (df[0]: complete; df[1]: packed under the MFMAs of q6): two 1 KiB stores
        // per wave, each issued behind an MFMA
        uint64_t tb = 0;
        if constexpr (DS) {
            const int sl_ = (int)(qs >> 5);
            const char *tile = ds_base + ((ds_row_base(sl_ >> 3, ds_nkb, ds_tri) + (sl_ & 7)) << 11); // DS_TILE = 2^11
            tb = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile) |
                 ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)tile >> 32)) << 32);
        }
        k4_wait_tr<8>(t6);
#define K4_Q6(DD, ID)                                                                                                   \
    dk[DD] = a_mfma<BF>(k4_frag<BF>(t6, DD), df[0], dk[DD]);                                                            \
    asm volatile("" : "+a"(dk[DD]), "+v"(dpv));                                                                         \
    K4_CVT2(df[1], 2 * (DD), dpv[8 + 2 * (DD)], dpv[9 + 2 * (DD)])                                                      \
    if constexpr (LAST && K4_SPREAD) stage_piece(std::integral_constant<int, ID>{}, pr + 3, (it + 3) & 3);              \
    /* write-through (sc0 sc1): the lines leave the XCD's L2 instead of evicting the Q / dO slices the workgroups of a head share */ \
    if (DS && (DD) == 1) asm volatile("global_store_dwordx4 %0, %1, %2 offset:0 sc0 sc1" : : "v"(ds_lane), "v"(df[0]), "s"(tb) : "memory"); \
    asm volatile("" : "+a"(dk[((DD) + 1) & 3]), "+v"(df[1]) : : "memory");                                              \
    k4_rowc1<NOFF + 32 * (DD)>(ln, cs[DD]);                                                                             \
    k4_row1a<NOFF + 512 * ((DD) >> 1)>(((DD) & 1) ? on : en, g0[DD]);
        K4_Q6(0, 0) K4_Q6(1, 1) K4_Q6(2, 2) K4_Q6(3, 3)
#undef K4_Q6
        TLK_STAMP(7)
        // q7: dK k-step 1
        k4_wait_tr<8>(t7);
#define K4_Q7(DD, ID)                                                                                                   \
    dk[DD] = a_mfma<BF>(k4_frag<BF>(t7, DD), df[1], dk[DD]);                                                            \
    if constexpr (LAST && K4_SPREAD) stage_piece(std::integral_constant<int, ID>{}, pr + 3, (it + 3) & 3);              \
    if (DS && (DD) == 1) asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 sc0 sc1" : : "v"(ds_lane), "v"(df[1]), "s"(tb) : "memory"); \
    asm volatile("" : "+a"(dk[((DD) + 1) & 3]) : : "memory");                                                           \
    k4_row1a<NOFF + 1024 + 512 * ((DD) >> 1)>(((DD) & 1) ? on : en, g1[DD]);                                            \
    k4_rowc1<NOFF + 128 + 32 * (DD)>(ln, cp[DD]);
        K4_Q7(0, 4) K4_Q7(1, 5) K4_Q7(2, 6) K4_Q7(3, 7)
#undef K4_Q7
        if constexpr (LAST && K4_SPREAD) {
            stage_piece(std::integral_constant<int, 8>{}, pr + 3, (it + 3) & 3);
            stage_piece(std::integral_constant<int, 9>{}, pr + 3, (it + 3) & 3);
        }
        TLK_STAMP(9)
#ifdef KF_ATTN_TIMELINE
        asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(tlk[0]), "+s"(tlk[1]), "+s"(tlk[2]), "+s"(tlk[3]), "+s"(tlk[4]), "+s"(tlk[5]), "+s"(tlk[6]), "+s"(tlk[7]),
                                              "+s"(tlk[8]), "+s"(tlk[9]) : : "memory");
        tlk_acc[0] += tlk[1] - tlk[0]; tlk_acc[1] += tlk[2] - tlk[1]; tlk_acc[2] += tlk[3] - tlk[2]; tlk_acc[3] += tlk[4] - tlk[3];
        tlk_acc[4] += tlk[5] - tlk[4]; tlk_acc[5] += tlk[6] - tlk[5]; tlk_acc[6] += tlk[8] - tlk[6]; tlk_acc[7] += tlk[7] - tlk[8];
        tlk_acc[8] += tlk[9] - tlk[7]; tlk_acc[9] += tlk[0] - tlk_prev; tlk_prev = tlk[9]; tlk_acc[10] += 1;
#endif
    };

    // Head size 64: the same slice with half the k-steps and half the column blocks - six phases of 4, 4, 2, 2, 2, 2 MFMAs
    // (S | dP | dV k0 | dV k1 | dK k0 | dK k1), written like the head-size-128 slice: asm score chains on VGPR accumulators that start
    // from the row constants, P and dS in place, two LDS reads behind every MFMA. The LDS images, offsets and the dS tiles are those of
    // head size 128 (column blocks 2 and 3 of a tile simply stay unused); 6 DMA operations per pair. A slice's 16 MFMAs carry the same
    // 16 scores per lane as 32 do at head size 128, so it is bound by its arithmetic: about 7 instructions per MFMA gap.
    // LDS reads in issue order (each group 4 reads, two per gap), and the quarter-phase that consumes them:
    //   p0: t4 (dO^T k0 -> p4), t5 (dO^T k1 -> p5) | p2: t6 (Q^T k0 -> p6), t7 (Q^T k1 -> p7) | p4: the next slice's S constants (-> its p0)
    //   | p5: its Q rows (-> p0) | p6: its dP constants (-> p2) | p7: its dO rows (-> p2; carried in g1)
    // counted waits (reads issued after the awaited group): p0 8, p2 8, p4 12, p5 12, p6 12, p7 12. The pair barrier sits in front of
    // p4: everything read after it belongs to the next pair's buffer when the slice is a pair's last.
    auto slice_body64 = [&, ds_lane, ds_base, ds_nkb, ds_tri](auto mask_c, auto soff_c, auto noff_c, auto last_c, unsigned e, unsigned o, unsigned t0,
                            unsigned t1, unsigned l, unsigned en, unsigned on, unsigned ln, int64_t qs, int pr, int it) __attribute__((always_inline)) {
        constexpr bool MASK = decltype(mask_c)::value, LAST = decltype(last_c)::value;
        constexpr int SOFF = decltype(soff_c)::value, NOFF = decltype(noff_c)::value, DO = SOFF + BQS * AROW, NDO = NOFF + BQS * AROW;
        f32x16 sv, dpv;
        const int nd = (int)(n - qs) - 4 * hl; // key - query of accumulator element e is nd - a_row(e, 0): masked when positive
        K4Tr t4, t5, t6, t7;
        frag_t pf[2], df[2];
#define K6_EXP(E)                                                        \
    {                                                                    \
        float pv = __builtin_amdgcn_exp2f(sv[E] * c);                    \
        if (MASK && nd > a_row(E, 0)) pv = 0.f;                          \
        if (KF_MUT(2, xb == 0 && qs == 4 * BQS)) pv = 0.f;               \
        sv[E] = pv;                                                      \
    }
#define K6_MUL2(E) dpv[E] = k4_mul(sv[E], dpv[E]); dpv[(E) + 1] = k4_mul_odd(sv[(E) + 1], dpv[(E) + 1]);
        // p0: S (4 MFMAs)
        k4_wait4a_cv<8>(g0, cs);
        f32x16 c_s = k4_acc(cs), c_p;
        k4_mfma_first<BF, true>(sv, g0[0], kf[0], c_s);
        k4_trp<DO>(t0, t1, t4.lo[0], t4.hi[0]);
        k4_mfma_acc<BF, true>(sv, g0[1], kf[1]);
        k4_trp<DO + 512>(t0, t1, t4.lo[1], t4.hi[1]);
        k4_mfma_acc<BF, true>(sv, g0[2], kf[2]);
        k4_trp<DO + 4096>(t0, t1, t5.lo[0], t5.hi[0]);
        k4_mfma_acc<BF, true>(sv, g0[3], kf[3]);
        k4_trp<DO + 4096 + 512>(t0, t1, t5.lo[1], t5.hi[1]);
        // p2: dP (4 MFMAs); p = exp2(S'') for elements 0..11, P of k-step 0 packed
        k4_wait4_cv<8>(g1, cp);
        c_p = k4_acc(cp);
        asm volatile("s_nop 3" : "+v"(sv)); // (S is read three instructions on: tools/kernel_hazards.py)
        k4_mfma_first_t<BF>(dpv, g1[0], vf[0], c_p, sv);
        K6_EXP(0) K6_EXP(1) K6_EXP(2)
        k4_trp<SOFF>(t0, t1, t6.lo[0], t6.hi[0]);
        k4_mfma_acc_t<BF>(dpv, g1[1], vf[1], sv);
        K6_EXP(3) K6_EXP(4) K6_EXP(5) K4_CVTP(pf[0], 0, sv[0], sv[1]) K4_CVTP(pf[0], 2, sv[2], sv[3])
        k4_trp<SOFF + 512>(t0, t1, t6.lo[1], t6.hi[1]);
        k4_mfma_acc_t<BF>(dpv, g1[2], vf[2], sv);
        K6_EXP(6) K6_EXP(7) K6_EXP(8) K4_CVTP(pf[0], 4, sv[4], sv[5])
        k4_trp<SOFF + 4096>(t0, t1, t7.lo[0], t7.hi[0]);
        k4_mfma_acc_t<BF>(dpv, g1[3], vf[3], sv);
        K6_EXP(9) K6_EXP(10) K6_EXP(11) K4_CVTP(pf[0], 6, sv[6], sv[7])
        k4_trp<SOFF + 4096 + 512>(t0, t1, t7.lo[1], t7.hi[1]);
        k4_keep(c_s);
        // p4: dV k-step 0 (2 MFMAs); the last exponentials; dS = p dP' for elements 0..7; P of k-step 1 packed
        if constexpr (LAST) {
            // as for head size 128 (see there): the next pair has landed, the previous pair's buffer is free for pair pr + 3. With 6 DMA
            // operations per pair and the two dS stores behind MFMAs of p6 / p7: 2 + (6 + 2) + 2 = 12 younger operations.
            if constexpr (DS) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
        }
        uint64_t tb = 0;
        if constexpr (DS) {
            const int sl_ = (int)(qs >> 5);
            const char *tile = ds_base + ((ds_row_base(sl_ >> 3, ds_nkb, ds_tri) + (sl_ & 7)) << 11);
            tb = (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)tile) |
                 ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)tile >> 32)) << 32);
        }
        k4_wait_tr2<12>(t4);
        asm volatile("s_nop 2" : "+v"(dpv)); // (the dP chain's result is read a few instructions on)
        dv[0] = a_mfma<BF>(k4_frag<BF>(t4, 0), pf[0], dv[0]);
        asm volatile("" : "+a"(dv[0]), "+v"(dpv), "+v"(sv));
        K6_EXP(12) K6_EXP(13) K6_MUL2(0) K6_MUL2(2) K4_CVTP(pf[1], 0, sv[8], sv[9]) K4_CVTP(pf[1], 2, sv[10], sv[11])
        asm volatile("" : "+a"(dv[1]), "+v"(dpv), "+v"(sv));
        k4_rowc1<NOFF>(ln, cs[0]); k4_rowc1<NOFF + 32>(ln, cs[1]);
        dv[1] = a_mfma<BF>(k4_frag<BF>(t4, 1), pf[0], dv[1]);
        asm volatile("" : "+a"(dv[1]), "+v"(dpv), "+v"(sv));
        K6_EXP(14) K6_EXP(15) K6_MUL2(4) K6_MUL2(6) K4_CVTP(pf[1], 4, sv[12], sv[13]) K4_CVTP(pf[1], 6, sv[14], sv[15])
        asm volatile("" : "+a"(dv[0]), "+v"(dpv), "+v"(pf[1]));
        k4_rowc1<NOFF + 64>(ln, cs[2]); k4_rowc1<NOFF + 96>(ln, cs[3]);
        k4_keep(c_p);
        k4_keep(g1);
        // p5: dV k-step 1; dS for elements 8..15; dS of k-step 0 packed
        k4_wait_tr2<12>(t5);
        dv[0] = a_mfma<BF>(k4_frag<BF>(t5, 0), pf[1], dv[0]);
        asm volatile("" : "+a"(dv[0]), "+v"(dpv));
        K6_MUL2(8) K6_MUL2(10) K4_CVT2(df[0], 0, dpv[0], dpv[1]) K4_CVT2(df[0], 2, dpv[2], dpv[3])
        asm volatile("" : "+a"(dv[1]), "+v"(dpv), "+v"(df[0]));
        k4_row1a<NOFF>(en, g0[0]); k4_row1a<NOFF>(on, g0[1]);
        dv[1] = a_mfma<BF>(k4_frag<BF>(t5, 1), pf[1], dv[1]);
        asm volatile("" : "+a"(dv[1]), "+v"(dpv));
        K6_MUL2(12) K6_MUL2(14) K4_CVT2(df[0], 4, dpv[4], dpv[5]) K4_CVT2(df[0], 6, dpv[6], dpv[7])
        asm volatile("" : "+a"(dk[0]), "+v"(dpv), "+v"(df[0]));
        k4_row1a<NOFF + 512>(en, g0[2]); k4_row1a<NOFF + 512>(on, g0[3]);
        // p6: dK k-step 0; dS of k-step 1 packed; first dS store
        k4_wait_tr2<12>(t6);
        dk[0] = a_mfma<BF>(k4_frag<BF>(t6, 0), df[0], dk[0]);
        asm volatile("" : "+a"(dk[0]), "+v"(dpv));
        K4_CVT2(df[1], 0, dpv[8], dpv[9]) K4_CVT2(df[1], 2, dpv[10], dpv[11])
        if constexpr (LAST) stage_piece(std::integral_constant<int, 0>{}, pr + 3, (it + 3) & 3);
        asm volatile("" : "+a"(dk[1]), "+v"(df[1]) : : "memory");
        k4_rowc1<NOFF + 128>(ln, cp[0]); k4_rowc1<NOFF + 160>(ln, cp[1]);
        dk[1] = a_mfma<BF>(k4_frag<BF>(t6, 1), df[0], dk[1]);
        asm volatile("" : "+a"(dk[1]), "+v"(dpv));
        K4_CVT2(df[1], 4, dpv[12], dpv[13]) K4_CVT2(df[1], 6, dpv[14], dpv[15])
        if constexpr (LAST) stage_piece(std::integral_constant<int, 1>{}, pr + 3, (it + 3) & 3);
        if constexpr (DS) asm volatile("global_store_dwordx4 %0, %1, %2 offset:0 sc0 sc1" : : "v"(ds_lane), "v"(df[0]), "s"(tb) : "memory");
        asm volatile("" : "+a"(dk[0]), "+v"(df[1]) : : "memory");
        k4_rowc1<NOFF + 192>(ln, cp[2]); k4_rowc1<NOFF + 224>(ln, cp[3]);
        // p7: dK k-step 1; second dS store; the next slice's dO rows
        k4_wait_tr2<12>(t7);
        dk[0] = a_mfma<BF>(k4_frag<BF>(t7, 0), df[1], dk[0]);
        asm volatile("" : "+a"(dk[0]));
        if constexpr (LAST) stage_piece(std::integral_constant<int, 4>{}, pr + 3, (it + 3) & 3);
        asm volatile("" : "+a"(dk[1]) : : "memory");
        k4_row1<NDO>(en, g1[0]); k4_row1<NDO>(on, g1[1]);
        dk[1] = a_mfma<BF>(k4_frag<BF>(t7, 1), df[1], dk[1]);
        asm volatile("" : "+a"(dk[1]));
        if constexpr (LAST) stage_piece(std::integral_constant<int, 5>{}, pr + 3, (it + 3) & 3);
        if constexpr (DS) asm volatile("global_store_dwordx4 %0, %1, %2 offset:1024 sc0 sc1" : : "v"(ds_lane), "v"(df[1]), "s"(tb) : "memory");
        asm volatile("" : "+a"(dk[0]) : : "memory");
        k4_row1<NDO + 512>(en, g1[2]); k4_row1<NDO + 512>(on, g1[3]);
        if constexpr (LAST) {
            stage_piece(std::integral_constant<int, 6>{}, pr + 3, (it + 3) & 3);
            stage_piece(std::integral_constant<int, 9>{}, pr + 3, (it + 3) & 3);
        }
#undef K6_EXP
#undef K6_MUL2
    };

    using I0 = std::integral_constant<int, 0>;
    using IS = std::integral_constant<int, K4SL>;
    auto pair_body = [&](auto mask_c, int pr, int it) __attribute__((always_inline)) {
        const unsigned bo = (unsigned)((it & 3) * K4PAIR);           // this pair's buffer
        const unsigned bn = (unsigned)(((it + 1) & 3) * K4PAIR);     // the next pair's
        const int64_t qa = (int64_t)pr * 2 * BQS, qb = qa + BQS;
        const unsigned e = rb_e + bo, o = rb_o + bo, t0 = tb_0 + bo, t1 = tb_1 + bo, l = lr + bo;
        if constexpr (D == 128) {
            slice_body(mask_c, I0{}, IS{}, std::false_type{}, e, o, t0, t1, l, e, o, l, qa, pr, it);
            slice_body(mask_c, IS{}, I0{}, std::true_type{}, e, o, t0, t1, l, rb_e + bn, rb_o + bn, lr + bn, qb, pr, it);
        } else {
            slice_body64(mask_c, I0{}, IS{}, std::false_type{}, e, o, t0, t1, l, e, o, l, qa, pr, it);
            slice_body64(mask_c, IS{}, I0{}, std::true_type{}, e, o, t0, t1, l, rb_e + bn, rb_o + bn, lr + bn, qb, pr, it);
        }
    };
    // two loops, one body each (a loop that switches between the masked and the plain body makes the allocator shuttle
    // the dK / dV accumulators between the two register files at every iteration)
    // pairs p0 .. pm - 1 touch this wave's diagonal (pair pr holds queries 64 pr .. 64 pr + 63; masked while 64 pr < kw + 31): a
    // plain scalar trip count (with the 64-bit compare in the loop condition a variant of this kernel had three accumulator tuples
    // parked in scratch at every iteration of the masked loop)
    int pm = (int)((kw + 31 + 2 * BQS - 1) / (2 * BQS));
    pm = __builtin_amdgcn_readfirstlane(pm < np ? pm : np);
    int pr = p0, it = 0;
#ifdef KF_ATTN_TIMELINE
    const unsigned long long tl_loop0 = __builtin_amdgcn_s_memtime();
#endif
    for (; pr < pm; ++pr, ++it) pair_body(std::true_type{}, pr, it);
    for (; pr < np; ++pr, ++it) pair_body(std::false_type{}, pr, it);
#ifdef KF_ATTN_TIMELINE
    const unsigned long long tl_loop1 = __builtin_amdgcn_s_memtime();
#endif
#undef K4_MFMA4
#undef K4_CVT2
#undef K4_CVTP
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); // drain the ring and the last prefetch before LDS is reused
    __syncthreads();
    // (Round 3, tried and dropped: requesting the NEXT pass's K / V fragments and first three slice pairs here, in front of this
    // epilogue - slabs moved to the ring's fourth slot - so that their latency runs under it: 2.32-2.35 ms against 2.29-2.31 on the
    // same box (tools/scratch/ab_attn.sh): thirty LDS-DMA issues in front of the stores delay the epilogue by more than the next
    // pass's prologue gains.)
    a_store_rows<BF, DB>(smem + wid * 32 * OPAD, a.dv + a_head(a.ldv, bh, a.H) + kw * a.ldv.sr, dv, BF ? 1.f : 1.f / kPShiftF16, a.ldv.sr);
    a_store_rows<BF, DB>(smem + wid * 32 * OPAD, a.dk + a_head(a.ldk, bh, a.H) + kw * a.ldk.sr, dk, a.scale, a.ldk.sr);
    if (a.persist) __syncthreads(); // the staging slabs overlap the ring the next block fills
#ifdef KF_ATTN_TIMELINE
    if (lane == 0 && g_attn_tl && D == 128) { // q0..q5, q6's wait + barrier, q6, q7, between slices, slices, prologue, epilogue
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long tl_end = __builtin_amdgcn_s_memtime();
        unsigned long long *dst = g_attn_tl + ((size_t)blockIdx.x * 2 + (pass & 1)) * 64 + wid * 16;
        for (int i = 0; i < 11; ++i) dst[i] = tlk_acc[i];
        dst[11] = tl_loop0 - tl_pass0;
        dst[12] = tl_end - tl_loop1;
    }
#endif
  }
}

// ------------------------------------------------------------------------------------------
// dK / dV, round 4: 4 waves x 64 keys, one wave per SIMD, the whole 512-register file asm-owned (dK^T and dV^T of the wave's 64 keys in
// all 256 accumulator registers, its K and V fragments in 128 vector registers), a block's pass as ONE generated instruction stream:
// tools/gen_attn_dkv.py -> attn_dkv_w4.inc (structure, register map, placement documented there). The wrapper maps the workgroup to
// its two 256-key blocks (a block and its causal mirror) and hands over scalars only. Head size 128, Skv a multiple of 256, K / V and
// dK / dV with one row stride each; everything else keeps attn_bwd_dkv_v4_kernel. Same dS tiles, same row constants.
// ------------------------------------------------------------------------------------------
#ifndef KF_DKV_W4_INC
#define KF_DKV_W4_INC "attn_dkv_w4.inc"
#endif
#include KF_DKV_W4_INC
constexpr int K5B = 256; // keys per block
// SQ: the scaled-K form (KF_ATTN_SCALED_OPERANDS; the row constants must then be -lse log2 e); default: exact f32 scores, row constants -lse / scale
// D64: the head-size-64 stream (round 5; exact scores only: SQ must be false)
template <bool BF, bool DS, bool SQ, bool D64 = false>
__global__ __launch_bounds__(256) void attn_bwd_dkv_w4_kernel(const AttnArgs a) {
    static_assert(!(SQ && D64), "the scaled-K form exists for head size 128 only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nkb = (int)((a.Skv + K5B - 1) / K5B), nwx = a.persist ? nkb / 2 : nkb;
    const unsigned lds = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned cdelta = (unsigned)((const char *)a.ndelta - (const char *)a.nlse);
    const int qsr = (int)a.lq.sr, dosr = (int)a.ldo.sr, kvsr = (int)a.lk.sr, osr = (int)a.ldk.sr;
    const float scale = SQ ? a.scale : a.scale_log2e; // (what the stream multiplies by: K once per block | every score in front of its exp2)
    const float scl = a.scale;                         // (dK = scale dS^T Q)
    const int ns_all = (int)((a.Sq + BQS - 1) / BQS);
    const int dbytes = D64 ? 128 : 256;
    // what the stream's descriptors may touch (bytes from their bases; gen_attn_dkv.py prologue): Q / dO rows below Sq, this block's K / V / dK / dV rows below Skv
    const unsigned qn = (unsigned)((a.Sq - 1) * a.lq.sr + dbytes), don = (unsigned)((a.Sq - 1) * a.ldo.sr + dbytes);
#ifdef KF_DKV_W4_STAMPS // (the diagnostic build clobbers 22 more scalar registers: no room for the loop's state beside the stream's inputs)
    {
        const unsigned vwg = blockIdx.x;
#else
#pragma nounroll
    for (unsigned vwg = blockIdx.x; vwg < a.nvwg; vwg += gridDim.x) { // (virtual workgroups: see attn_fwd_w4_kernel)
#endif
    int xb0;
    int64_t bh;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh, vwg);
    bh += a.bh0;
    const char *qp = a.q + a_head(a.lq, bh, a.H), *dop = a.d_o + a_head(a.ldo, bh, a.H);
    const char *kh = a.k + a_head(a.lk, bh, a.H), *vh = a.v + a_head(a.lv, bh, a.H);
    char *dkh = a.dk + a_head(a.ldk, bh, a.H), *dvh = a.dv + a_head(a.ldv, bh, a.H);
    const float *cp = a.nlse + bh * a.Sqc;
#pragma nounroll
    for (int pass = 0; pass < (a.persist ? 2 : 1); ++pass) {
        const int xb = ((pass & 1) != (a.persist_rev != 0)) ? nkb - 1 - xb0 : xb0;
        const int64_t k0 = (int64_t)xb * K5B;
        const char *kp = kh + k0 * a.lk.sr, *vp = vh + k0 * a.lv.sr;
        char *dkp = dkh + k0 * a.ldk.sr, *dvp = dvh + k0 * a.ldv.sr;
        const int64_t krows = a.Skv - k0 < K5B ? a.Skv - k0 : K5B; // the block's keys that exist
        const unsigned kn = (unsigned)((krows - 1) * a.lk.sr + dbytes), on = (unsigned)((krows - 1) * a.ldk.sr + dbytes);
        const int s0 = (int)(k0 / BQS);                    // the first slice with a query that sees one of the block's keys
        const int ns = ns_all;                             // (a block beyond the last query, s0 >= ns: no slices, zero gradients; the stream clamps its first requests to slice ns - 1)
        // this block's first square of the dS workspace (ds_tile_index): row of query block xb, square xb; the stream walks down from there - dsrs =
        // the bytes from a query block's last slice to the next block's first (its row's length less seven tiles), growing by a square per row up to dsrm
        const char *dsp = DS ? a.ds + ((bh - a.bh0) * a.ds_pair + ds_row_base(xb, a.ds_nkb, a.ds_tri) + (int64_t)xb * 64) * DS_TILE : nullptr;
        const int64_t rowlen = (a.ds_tri && xb + 1 < a.ds_nkb) ? xb + 1 : a.ds_nkb;   // (full rows: the step never grows - it starts at its maximum)
        const unsigned dsrs = (unsigned)((rowlen * 64 - 7) * DS_TILE), dsrm = (unsigned)((a.ds_nkb * 64 - 7) * DS_TILE);
        int mut = -1;
#ifdef KF_MUTANT
        if (a.mutant == 2 && xb == 0 && wid < 2) mut = 4; // defect 2: queries 128..159 contribute nothing to keys 0..127 (a steady slice of waves 0 and 1)
#endif
#ifdef KF_DKV_W4_STAMPS // diagnostic build (tools/attn_dkv_w4_timeline.py): eight cycle sums per wave and block pass into the debug buffer
        const char *dbg = (const char *)a.dbg + (size_t)(vwg * 2 + pass) * 4 * 64;
#define KF_DKV_EXTRA , [dbg] "s"(dbg)
        // (the diagnostic stream clobbers 22 more scalar registers: the four descriptor bounds become the literal "no bound" - whole tiles only)
        (void)qn; (void)don; (void)kn; (void)on;
#define KF_DKV_BOUNDS [qn] "n"(-1), [don] "n"(-1), [kn] "n"(-1), [on] "n"(-1)
#else
#define KF_DKV_EXTRA
#define KF_DKV_BOUNDS [qn] "s"(qn), [don] "s"(don), [kn] "s"(kn), [on] "s"(on)
#endif
#define KF_DKV_OPERANDS                                                                                                                       \
    [qp] "s"(qp), [dop] "s"(dop), [kp] "s"(kp), [vp] "s"(vp), [dkp] "s"(dkp), [dvp] "s"(dvp), [cp] "s"(cp), [dsp] "s"(dsp), [cdelta] "s"(cdelta), \
        [qsr] "s"(qsr), [dosr] "s"(dosr), [kvsr] "s"(kvsr), [osr] "s"(osr), [s0] "s"(s0), [ns] "s"(ns), [dsrs] "s"(dsrs), [dsrm] "s"(dsrm), [wid] "s"(wid),       \
        [scale] "s"(scale), [scl] "s"(scl), [lds] "s"(lds), [mut] "s"(mut), KF_DKV_BOUNDS KF_DKV_EXTRA
#define KF_DKV_ASM(TEXT) asm volatile(TEXT : : KF_DKV_OPERANDS : KF_DKV_W4_CLOBBERS)
        if constexpr (D64) {
            if constexpr (BF && DS) KF_DKV_ASM(KF_DKV_W4_D64_ASM_BF16_DS);
            else if constexpr (BF) KF_DKV_ASM(KF_DKV_W4_D64_ASM_BF16_NODS);
            else if constexpr (DS) KF_DKV_ASM(KF_DKV_W4_D64_ASM_F16_DS);
            else KF_DKV_ASM(KF_DKV_W4_D64_ASM_F16_NODS);
        } else if constexpr (SQ) {
            if constexpr (BF && DS) KF_DKV_ASM(KF_DKV_W4_ASM_BF16_DS_SQ);
            else if constexpr (BF) KF_DKV_ASM(KF_DKV_W4_ASM_BF16_NODS_SQ);
            else if constexpr (DS) KF_DKV_ASM(KF_DKV_W4_ASM_F16_DS_SQ);
            else KF_DKV_ASM(KF_DKV_W4_ASM_F16_NODS_SQ);
        } else {
            if constexpr (BF && DS) KF_DKV_ASM(KF_DKV_W4_ASM_BF16_DS);
            else if constexpr (BF) KF_DKV_ASM(KF_DKV_W4_ASM_BF16_NODS);
            else if constexpr (DS) KF_DKV_ASM(KF_DKV_W4_ASM_F16_DS);
            else KF_DKV_ASM(KF_DKV_W4_ASM_F16_NODS);
        }
#undef KF_DKV_ASM
#undef KF_DKV_OPERANDS
#undef KF_DKV_EXTRA
#undef KF_DKV_BOUNDS
    }
    }
}

// ==========================================================================================
// forward, f32 (the reference's dtype: CausalAttentionForwardFN<float, 64 | 128>, causal_attention.h:66-258), on the
// exact-f32 matrix instruction v_mfma_f32_32x32x2_f32 (a k-ordered fma chain, 157 TFLOP/s dense, 1/16 of the bf16 rate -
// so this kernel is matrix-pipe bound by a wide margin and needs no scheduling tricks). Query on the lane, 4 waves x 32
// queries, 32-key K / V tiles in LDS (rows padded by 4 floats: conflict-free ds_read_b128 row reads).
//   S^T = K Q^T: A = K[key][k] from LDS, B = Q[query][k] from registers; lane half hl owns k in [hl D/2, (hl+1) D/2) of
//   BOTH operands (any consistent assignment of k to the two halves is a valid MFMA), so every LDS read is 16 B of one row.
//   O^T += V^T P^T: step e uses the two keys row(e, 0), row(e, 1) the S accumulator register e holds for the two lane
//   halves - P is consumed straight out of the accumulator, no shuffle, no conversion; A = V[row(e, hl)][d] from LDS.
// ==========================================================================================
constexpr int XQ = 128, XK = 32; // queries per block, keys per tile

template <int D>
__global__ __launch_bounds__(256, 2) void attn_fwd_f32_mfma_kernel(const AttnArgs a) {
    constexpr int DP = D + 4, HD = D / 2, ND = D / 32; // padded row (floats); k per lane half; 32-column blocks of O
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ks = (float *)smem, *Vs = Ks + XK * DP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, xl = lane & 31, hl = lane >> 5, t = threadIdx.x;
    int xb0;
    int64_t bh;
    const int nxb = (int)((a.Sq + XQ - 1) / XQ);
    const int nwx = a.persist ? nxb / (2 * a.persist) : nxb; // a.persist: a block and its causal mirror per workgroup (see attn_fwd_v3_kernel)
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    const float *Kg = (const float *)a.k + bh * a.Skv * D;
    const float *Vg = (const float *)a.v + bh * a.Skv * D;
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = (pass & 1) ? nxb - 1 - xp : xp;
    const int qblk = nxb - 1 - xb; // longest blocks first
    const int64_t q0 = (int64_t)qblk * XQ, qw = q0 + wid * 32, m = qw + xl;
    const bool active = qw < a.Sq; // Sq % 32 == 0: whole waves

    float qreg[HD];
    if (active) {
        const float4 *Qg = (const float4 *)((const float *)a.q + (bh * a.Sq + m) * D + hl * HD);
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 v = Qg[j];
            qreg[4 * j] = v.x; qreg[4 * j + 1] = v.y; qreg[4 * j + 2] = v.z; qreg[4 * j + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < HD; ++j) qreg[j] = 0.f;
    }
    f32x16 o[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[d][e] = 0.f;
    float m_i = -INFINITY, l_i = 0.f;
    const float c = a.scale * kLog2e;

    const int64_t q_end = q0 + XQ < a.Sq ? q0 + XQ : a.Sq;
    const int64_t kv_end = a.Skv < q_end ? a.Skv : q_end;
    const int nt = (int)((kv_end + XK - 1) / XK);
    // tile staging: XK x D floats = XK * D / 4 float4, 256 threads
    constexpr int NV = XK * D / 4 / 256; // float4 per thread per tile (4 at D = 128, 2 at D = 64)
    // named registers, not arrays: hipcc keeps float4[N] staging arrays that cross a loop in scratch
    float4 k0, k1, k2, k3, v0, v1, v2, v3;
    k2 = k3 = v2 = v3 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto src = [&](const float *g, int tile, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (const float4 *)(g + ((int64_t)tile * XK + row) * D + 4 * c4);
    };
    auto dst = [&](float *l, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (float4 *)(l + row * DP + 4 * c4);
    };
    auto gload = [&](int tile) __attribute__((always_inline)) {
        k0 = *src(Kg, tile, 0); k1 = *src(Kg, tile, 1); v0 = *src(Vg, tile, 0); v1 = *src(Vg, tile, 1);
        if constexpr (NV == 4) { k2 = *src(Kg, tile, 2); k3 = *src(Kg, tile, 3); v2 = *src(Vg, tile, 2); v3 = *src(Vg, tile, 3); }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
        *dst(Ks, 0) = k0; *dst(Ks, 1) = k1; *dst(Vs, 0) = v0; *dst(Vs, 1) = v1;
        if constexpr (NV == 4) { *dst(Ks, 2) = k2; *dst(Ks, 3) = k3; *dst(Vs, 2) = v2; *dst(Vs, 3) = v3; }
    };
    gload(0);
    for (int tl = 0; tl < nt; ++tl) {
        const int64_t kv0 = (int64_t)tl * XK;
        __syncthreads(); // every wave is done with the previous tile
        lstore();
        __syncthreads();
        if (tl + 1 < nt) gload(tl + 1); // in flight under this tile's 128 MFMAs
        if (!active || kv0 > qw + 31) continue;
        // ---- S^T = K Q^T
        f32x16 sv;
#pragma unroll
        for (int e = 0; e < 16; ++e) sv[e] = 0.f;
        const float *krow = Ks + xl * DP + hl * HD;
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 kk = *(const float4 *)(krow + 4 * j);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[4 * j], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[4 * j + 1], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[4 * j + 2], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[4 * j + 3], sv, 0, 0, 0);
        }
        // ---- online softmax (one query per lane; the other lane half holds the other 16 keys)
        const bool need_mask = kv0 + XK - 1 > qw;
        float mx = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (need_mask && kv0 + a_row(e, hl) > m) sv[e] = -INFINITY;
            mx = fmaxf(mx, sv[e]);
        }
        mx = a_half_max(mx);
        const float m_new = fmaxf(m_i, mx);
        const float mc = m_new * c;
        const float alpha = __builtin_amdgcn_exp2f(m_i * c - mc);
        float rs = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], c, -mc));
            sv[e] = p;
            rs += p;
        }
        rs = a_half_sum(rs);
        l_i = l_i * alpha + rs;
        m_i = m_new;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[d][e] *= alpha;
        // ---- O^T += V^T P^T
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float *vrow = Vs + a_row(e, hl) * DP + xl;
#pragma unroll
            for (int d = 0; d < ND; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[32 * d], sv[e], o[d], 0, 0, 0);
        }
    }
    // ---- epilogue: O^T tiles -> rows through LDS (reuses the tile buffers), 16-byte coalesced stores
    __syncthreads();
    float *slab = (float *)smem + wid * 32 * (D + 4); // [32 queries][D + 4]
    if (active) {
        const float inv = 1.f / l_i;
#pragma unroll
        for (int d = 0; d < ND; ++d)
#pragma unroll
            for (int e = 0; e < 16; ++e) slab[xl * (D + 4) + 32 * d + a_row(e, hl)] = o[d][e] * inv;
        // the wave reads back what it wrote: LDS operations of one wave complete in order
        float *og = (float *)a.out + (bh * a.Sq + qw) * D;
#pragma unroll
        for (int i = 0; i < 32 * D / 4 / 64; ++i) {
            const int id = lane + 64 * i, row = id / (D / 4), c4 = id % (D / 4);
            *(float4 *)(og + (int64_t)row * D + 4 * c4) = *(const float4 *)(slab + row * (D + 4) + 4 * c4);
        }
        if (a.lse && hl == 0) a.lse[bh * a.Sq + m] = (m_i * c + __builtin_amdgcn_logf(l_i)) * kLn2;
    }
    // (the next block's first __syncthreads() comes before its first tile store: the slabs are read by then)
  }
}

// ==========================================================================================
// backward, f32, on v_mfma_f32_32x32x2_f32 (no reference counterpart: the reference has no backward at all). Same
// orientation tricks as the f32 forward: every product keeps "the other index" on the lane, so P and dS are consumed
// straight out of accumulator registers as the B operand of the next product.
//   dQ kernel (query on the lane, 4 waves x 32 queries, 32-key K / V tiles in LDS):
//       S^T = K Q^T,  dP^T = V dO^T   (A = K / V rows from LDS, B = Q / dO registers; lane half hl owns k in [hl D/2, ..))
//       dS^T = P^T o (dP^T - delta);   dQ^T += K^T dS^T  (A = K[key row(e, hl)][d] from LDS, B = dS register e)
//   dK/dV kernel (key on the lane, 4 waves x 32 keys, 32-query Q / dO tiles + their lse / delta in LDS; one wave per SIMD:
//   128 accumulator + 128 operand registers):
//       S = Q K^T,  dP = dO V^T  (A = Q / dO rows from LDS, B = K / V registers);  P = exp2(c S - lse),  dS = P o (dP - delta)
//       dV^T += dO^T P  (A = dO[query row(e, hl)][d]),   dK^T += Q^T dS  (A = Q[query row(e, hl)][d])
// Exact-f32 fma chains throughout; deterministic (no atomics); causal pairing as in the other kernels.
// ==========================================================================================
template <int D>
__device__ __forceinline__ void x_store_rows(float *slab, float *dst, const f32x16 *acc, float mul) {
    // acc[d-block][e] = X^T[32 d + row(e, hl)][row-on-lane xl] -> 32 rows of D floats, 16-byte coalesced stores
    const int lane = threadIdx.x & 63, xl = lane & 31, hl = lane >> 5;
#pragma unroll
    for (int d = 0; d < D / 32; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[xl * (D + 4) + 32 * d + a_row(e, hl)] = acc[d][e] * mul;
#pragma unroll
    for (int i = 0; i < 32 * D / 4 / 64; ++i) {
        const int id = lane + 64 * i, row = id / (D / 4), c4 = id % (D / 4);
        *(float4 *)(dst + (int64_t)row * D + 4 * c4) = *(const float4 *)(slab + row * (D + 4) + 4 * c4);
    }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dq_f32_mfma_kernel(const AttnArgs a) {
    constexpr int DP = D + 4, HD = D / 2, ND = D / 32, NV = XK * D / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Ks = (float *)smem, *Vs = Ks + XK * DP;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, xl = lane & 31, hl = lane >> 5, t = threadIdx.x;
    int xb0;
    int64_t bh;
    const int nxb = (int)((a.Sq + XQ - 1) / XQ);
    const int nwx = a.persist ? nxb / (2 * a.persist) : nxb;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    const float *Kg = (const float *)a.k + bh * a.Skv * D;
    const float *Vg = (const float *)a.v + bh * a.Skv * D;
    const float c = a.scale * kLog2e;
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = (pass & 1) ? nxb - 1 - xp : xp;
    const int qblk = nxb - 1 - xb;
    const int64_t q0 = (int64_t)qblk * XQ, qw = q0 + wid * 32, m = qw + xl;
    const bool active = qw < a.Sq;
    float qreg[HD], doreg[HD];
    float lse2 = 0.f, dlt = 0.f;
    if (active) {
        const float4 *Qg = (const float4 *)((const float *)a.q + (bh * a.Sq + m) * D + hl * HD);
        const float4 *Og = (const float4 *)((const float *)a.d_o + (bh * a.Sq + m) * D + hl * HD);
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 v = Qg[j], w = Og[j];
            qreg[4 * j] = v.x; qreg[4 * j + 1] = v.y; qreg[4 * j + 2] = v.z; qreg[4 * j + 3] = v.w;
            doreg[4 * j] = w.x; doreg[4 * j + 1] = w.y; doreg[4 * j + 2] = w.z; doreg[4 * j + 3] = w.w;
        }
        lse2 = a.lse_r[bh * a.Sq + m] * kLog2e;
        dlt = a.delta[bh * a.Sq + m];
    } else {
#pragma unroll
        for (int j = 0; j < HD; ++j) { qreg[j] = 0.f; doreg[j] = 0.f; }
    }
    f32x16 dq[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) dq[d][e] = 0.f;
    const int64_t q_end = q0 + XQ < a.Sq ? q0 + XQ : a.Sq;
    const int64_t kv_end = a.Skv < q_end ? a.Skv : q_end;
    const int nt = (int)((kv_end + XK - 1) / XK);
    float4 k0, k1, k2, k3, v0, v1, v2, v3;
    k2 = k3 = v2 = v3 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto src = [&](const float *g, int tile, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (const float4 *)(g + ((int64_t)tile * XK + row) * D + 4 * c4);
    };
    auto dst = [&](float *l, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (float4 *)(l + row * DP + 4 * c4);
    };
    auto gload = [&](int tile) __attribute__((always_inline)) {
        k0 = *src(Kg, tile, 0); k1 = *src(Kg, tile, 1); v0 = *src(Vg, tile, 0); v1 = *src(Vg, tile, 1);
        if constexpr (NV == 4) { k2 = *src(Kg, tile, 2); k3 = *src(Kg, tile, 3); v2 = *src(Vg, tile, 2); v3 = *src(Vg, tile, 3); }
    };
    auto lstore = [&]() __attribute__((always_inline)) {
        *dst(Ks, 0) = k0; *dst(Ks, 1) = k1; *dst(Vs, 0) = v0; *dst(Vs, 1) = v1;
        if constexpr (NV == 4) { *dst(Ks, 2) = k2; *dst(Ks, 3) = k3; *dst(Vs, 2) = v2; *dst(Vs, 3) = v3; }
    };
    gload(0);
    for (int tl = 0; tl < nt; ++tl) {
        const int64_t kv0 = (int64_t)tl * XK;
        __syncthreads();
        lstore();
        __syncthreads();
        if (tl + 1 < nt) gload(tl + 1);
        if (!active || kv0 > qw + 31) continue;
        f32x16 sv, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sv[e] = 0.f; dp[e] = 0.f; }
        const float *krow = Ks + xl * DP + hl * HD, *vrow = Vs + xl * DP + hl * HD;
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 kk = *(const float4 *)(krow + 4 * j);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.x, qreg[4 * j], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.y, qreg[4 * j + 1], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.z, qreg[4 * j + 2], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(kk.w, qreg[4 * j + 3], sv, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 vv = *(const float4 *)(vrow + 4 * j);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.x, doreg[4 * j], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.y, doreg[4 * j + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.z, doreg[4 * j + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv.w, doreg[4 * j + 3], dp, 0, 0, 0);
        }
        const bool need_mask = kv0 + XK - 1 > qw;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], c, -lse2));
            if (need_mask && kv0 + a_row(e, hl) > m) p = 0.f;
            sv[e] = p * (dp[e] - dlt);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float *kr = Ks + a_row(e, hl) * DP + xl;
#pragma unroll
            for (int d = 0; d < ND; ++d) dq[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[32 * d], sv[e], dq[d], 0, 0, 0);
        }
    }
    __syncthreads();
    if (active) x_store_rows<D>((float *)smem + wid * 32 * (D + 4), (float *)a.dq + (bh * a.Sq + qw) * D, dq, a.scale);
    // (the next block's first __syncthreads() comes before its first tile store: the slabs are read by then)
  }
}

template <int D>
__global__ __launch_bounds__(256) void attn_bwd_dkv_f32_mfma_kernel(const AttnArgs a) {
    constexpr int DP = D + 4, HD = D / 2, ND = D / 32, NV = XK * D / 4 / 256;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float *Qs = (float *)smem, *Os = Qs + XK * DP, *Ls = Os + XK * DP; // Q tile | dO tile | lse2[32] | delta[32]
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, xl = lane & 31, hl = lane >> 5, t = threadIdx.x;
    int xb0;
    int64_t bh;
    const int nkb = (int)((a.Skv + XQ - 1) / XQ);
    const int nwx = a.persist ? nkb / (2 * a.persist) : nkb;
    a_block_map(nwx, a.nbh, a.xcd_map, xb0, bh);
    const float *Qg = (const float *)a.q + bh * a.Sq * D;
    const float *Og = (const float *)a.d_o + bh * a.Sq * D;
    const float c = a.scale * kLog2e;
#pragma nounroll
  for (int pass = 0; pass < (a.persist ? 2 * a.persist : 1); ++pass) {
    const int xp = xb0 + (pass >> 1) * nwx;
    const int xb = (pass & 1) ? xp : nkb - 1 - xp; // the short block (late keys) first
    const int64_t k0b = (int64_t)xb * XQ, kw = k0b + wid * 32, n = kw + xl;
    const bool active = kw < a.Skv;
    float kreg[HD], vreg[HD];
    if (active) {
        const float4 *Kp = (const float4 *)((const float *)a.k + (bh * a.Skv + n) * D + hl * HD);
        const float4 *Vp = (const float4 *)((const float *)a.v + (bh * a.Skv + n) * D + hl * HD);
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 v = Kp[j], w = Vp[j];
            kreg[4 * j] = v.x; kreg[4 * j + 1] = v.y; kreg[4 * j + 2] = v.z; kreg[4 * j + 3] = v.w;
            vreg[4 * j] = w.x; vreg[4 * j + 1] = w.y; vreg[4 * j + 2] = w.z; vreg[4 * j + 3] = w.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < HD; ++j) { kreg[j] = 0.f; vreg[j] = 0.f; }
    }
    f32x16 dk[ND], dv[ND];
#pragma unroll
    for (int d = 0; d < ND; ++d)
#pragma unroll
        for (int e = 0; e < 16; ++e) { dk[d][e] = 0.f; dv[d][e] = 0.f; }
    const int t0 = (int)(k0b / XK), nt = (int)((a.Sq + XK - 1) / XK); // query tiles holding a query >= the block's first key
    float4 q0r, q1r, q2r, q3r, o0r, o1r, o2r, o3r;
    q2r = q3r = o2r = o3r = make_float4(0.f, 0.f, 0.f, 0.f);
    float rc = 0.f; // threads 0..31: lse2 of the tile's query t, threads 32..63: its delta
    auto src = [&](const float *g, int tile, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (const float4 *)(g + ((int64_t)tile * XK + row) * D + 4 * c4);
    };
    auto dst = [&](float *l, int i) __attribute__((always_inline)) {
        const int id = t + 256 * i, row = id / (D / 4), c4 = id % (D / 4);
        return (float4 *)(l + row * DP + 4 * c4);
    };
    auto gload = [&](int tile) __attribute__((always_inline)) {
        q0r = *src(Qg, tile, 0); q1r = *src(Qg, tile, 1); o0r = *src(Og, tile, 0); o1r = *src(Og, tile, 1);
        if constexpr (NV == 4) { q2r = *src(Qg, tile, 2); q3r = *src(Qg, tile, 3); o2r = *src(Og, tile, 2); o3r = *src(Og, tile, 3); }
        if (t < 32) rc = a.lse_r[bh * a.Sq + (int64_t)tile * XK + t] * kLog2e;
        else if (t < 64) rc = a.delta[bh * a.Sq + (int64_t)tile * XK + t - 32];
    };
    auto lstore = [&]() __attribute__((always_inline)) {
        *dst(Qs, 0) = q0r; *dst(Qs, 1) = q1r; *dst(Os, 0) = o0r; *dst(Os, 1) = o1r;
        if constexpr (NV == 4) { *dst(Qs, 2) = q2r; *dst(Qs, 3) = q3r; *dst(Os, 2) = o2r; *dst(Os, 3) = o3r; }
        if (t < 64) Ls[t] = rc;
    };
    if (t0 < nt) gload(t0);
    for (int tl = t0; tl < nt; ++tl) {
        const int64_t qt0 = (int64_t)tl * XK;
        __syncthreads();
        lstore();
        __syncthreads();
        if (tl + 1 < nt) gload(tl + 1);
        if (!active || qt0 + 31 < kw) continue; // every query of the tile is before this wave's first key
        f32x16 sv, dp;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sv[e] = 0.f; dp[e] = 0.f; }
        const float *qrow = Qs + xl * DP + hl * HD, *orow = Os + xl * DP + hl * HD;
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 qq = *(const float4 *)(qrow + 4 * j);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(qq.x, kreg[4 * j], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(qq.y, kreg[4 * j + 1], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(qq.z, kreg[4 * j + 2], sv, 0, 0, 0);
            sv = __builtin_amdgcn_mfma_f32_32x32x2f32(qq.w, kreg[4 * j + 3], sv, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < HD / 4; ++j) {
            const float4 oo = *(const float4 *)(orow + 4 * j);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(oo.x, vreg[4 * j], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(oo.y, vreg[4 * j + 1], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(oo.z, vreg[4 * j + 2], dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_32x32x2f32(oo.w, vreg[4 * j + 3], dp, 0, 0, 0);
        }
        const bool need_mask = qt0 < kw + 31; // some query of the tile is before some key of the wave
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int r = a_row(e, hl);
            float p = __builtin_amdgcn_exp2f(__builtin_fmaf(sv[e], c, -Ls[r]));
            if (need_mask && n > qt0 + r) p = 0.f;
            sv[e] = p;
            dp[e] = p * (dp[e] - Ls[32 + r]);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float *orw = Os + a_row(e, hl) * DP + xl, *qrw = Qs + a_row(e, hl) * DP + xl;
#pragma unroll
            for (int d = 0; d < ND; ++d) {
                dv[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(orw[32 * d], sv[e], dv[d], 0, 0, 0);
                dk[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(qrw[32 * d], dp[e], dk[d], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    if (active) {
        float *slab = (float *)smem + wid * 32 * (D + 4);
        x_store_rows<D>(slab, (float *)a.dv + (bh * a.Skv + kw) * D, dv, 1.f);
        x_store_rows<D>(slab, (float *)a.dk + (bh * a.Skv + kw) * D, dk, a.scale);
    }
  }
}

// ==========================================================================================
// generic path: f32 math on the vector ALU, any Sq / Skv, D <= 256, f32 / bf16 / f16 storage.
// One block = 16 queries; key tiles of 32; 256 threads.
// ==========================================================================================
constexpr int GQ = 16, GK = 32;

template <typename T> __device__ __forceinline__ float t_load(const T *p) { return (float)*p; }
template <> __device__ __forceinline__ float t_load<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float t_load<f16_t>(const f16_t *p) { return f16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void t_store(T *p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void t_store<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void t_store<f16_t>(f16_t *p, float v) { *p = f32_to_f16(v); }

template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_generic_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = (int)a.D, DP = D + 1;
    float *Qs = (float *)smem;       // [GQ][DP]
    float *Ks = Qs + GQ * DP;        // [GK][DP]
    float *Vs = Ks + GK * DP;        // [GK][D]
    float *Ps = Vs + GK * D;         // [GQ][GK + 1]
    float *al = Ps + GQ * (GK + 1);  // [GQ] rescale factor of the current tile
    const int64_t bh = blockIdx.y, q0 = (int64_t)blockIdx.x * GQ;
    const T *Qg = (const T *)a.q + bh * a.Sq * D;
    const T *Kg = (const T *)a.k + bh * a.Skv * D;
    const T *Vg = (const T *)a.v + bh * a.Skv * D;
    const int t = threadIdx.x;

    for (int i = t; i < GQ * D; i += 256) {
        const int r = i / D, d = i % D;
        Qs[r * DP + d] = q0 + r < a.Sq ? t_load<T>(Qg + (q0 + r) * D + d) : 0.f;
    }
    const int orow = t / 16, ocol = t % 16; // output ownership: row orow, columns ocol + 16 j
    float o[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) o[j] = 0.f;
    float m_i = -INFINITY, l_i = 0.f; // replicated over the 16 lanes of a row

    const int64_t q_hi = q0 + GQ - 1 < a.Sq - 1 ? q0 + GQ - 1 : a.Sq - 1;
    const int64_t kv_end = a.Skv < q_hi + 1 ? a.Skv : q_hi + 1;
    for (int64_t kv0 = 0; kv0 < kv_end; kv0 += GK) {
        __syncthreads();
        for (int i = t; i < GK * D; i += 256) {
            const int r = i / D, d = i % D;
            const bool ok = kv0 + r < a.Skv;
            Ks[r * DP + d] = ok ? t_load<T>(Kg + (kv0 + r) * D + d) : 0.f;
            Vs[r * D + d] = ok ? t_load<T>(Vg + (kv0 + r) * D + d) : 0.f;
        }
        __syncthreads();
        // scores: thread -> (query t/32 and +8, key t%32)
        {
            const int kk = t % GK;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int qq = t / GK + 8 * h;
                float acc = 0.f;
                for (int d = 0; d < D; ++d) acc = fmaf(Qs[qq * DP + d], Ks[kk * DP + d], acc);
                acc *= a.scale;
                const int64_t mq = q0 + qq, nk = kv0 + kk;
                if (nk > mq || nk >= a.Skv) acc = -INFINITY;
                Ps[qq * (GK + 1) + kk] = acc;
            }
        }
        __syncthreads();
        // online softmax: 16 lanes per row, 2 keys each
        {
            float s0 = Ps[orow * (GK + 1) + ocol], s1 = Ps[orow * (GK + 1) + ocol + 16];
            float mx = fmaxf(s0, s1);
            for (int msk = 8; msk > 0; msk >>= 1) mx = fmaxf(mx, __shfl_xor(mx, msk, 64));
            const float m_new = fmaxf(m_i, mx);
            const float p0 = expf(s0 - m_new), p1 = expf(s1 - m_new);
            float rs = p0 + p1;
            for (int msk = 8; msk > 0; msk >>= 1) rs += __shfl_xor(rs, msk, 64);
            const float alpha = expf(m_i - m_new);
            l_i = l_i * alpha + rs;
            m_i = m_new;
            Ps[orow * (GK + 1) + ocol] = p0;
            Ps[orow * (GK + 1) + ocol + 16] = p1;
            if (ocol == 0) al[orow] = alpha;
        }
        __syncthreads();
        {
            const float alpha = al[orow];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int d = ocol + 16 * j;
                if (d < D) {
                    float acc = o[j] * alpha;
#pragma unroll 4
                    for (int kk = 0; kk < GK; ++kk) acc = fmaf(Ps[orow * (GK + 1) + kk], Vs[kk * D + d], acc);
                    o[j] = acc;
                }
            }
        }
    }
    const int64_t mq = q0 + orow;
    if (mq < a.Sq) {
        T *Og = (T *)a.out + (bh * a.Sq + mq) * D;
        const float inv = 1.f / l_i;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = ocol + 16 * j;
            if (d < D) t_store<T>(Og + d, o[j] * inv);
        }
        if (a.lse && ocol == 0) a.lse[bh * a.Sq + mq] = m_i + logf(l_i);
    }
}

// delta for the generic path: one wave per row
template <typename T>
__global__ __launch_bounds__(256) void attn_delta_generic_kernel(const T *o, const T *d_o, float *delta, int64_t nrows, int D) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    if (row < nrows)
        for (int d = lane; d < D; d += 64) acc += t_load<T>(o + row * D + d) * t_load<T>(d_o + row * D + d);
    for (int msk = 32; msk > 0; msk >>= 1) acc += __shfl_xor(acc, msk, 64);
    if (row < nrows && lane == 0) delta[row] = acc;
}

// generic backward, MODE 0: dQ for a block of 16 queries; MODE 1: dK and dV for a block of 16 keys.
// The owned block's rows play the role of "row", the swept tiles of 32 the role of "col".
template <typename T, int MODE>
__global__ __launch_bounds__(256) void attn_bwd_generic_kernel(const AttnArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = (int)a.D, DP = D + 1;
    float *Rq = (float *)smem;        // MODE 0: Q block   | MODE 1: K block     [GQ][DP]
    float *Rd = Rq + GQ * DP;         // MODE 0: dO block  | MODE 1: V block     [GQ][DP]
    float *Cq = Rd + GQ * DP;         // MODE 0: K tile    | MODE 1: Q tile      [GK][DP]
    float *Cd = Cq + GK * DP;         // MODE 0: V tile    | MODE 1: dO tile     [GK][DP]
    float *Ps = Cd + GK * DP;         // p   [GQ][GK + 1]
    float *Ds = Ps + GQ * (GK + 1);   // dS  [GQ][GK + 1]
    float *cl = Ds + GQ * (GK + 1);   // per-col lse (MODE 1) [GK]
    float *cd = cl + GK;              // per-col delta (MODE 1) [GK]
    const int64_t bh = blockIdx.y, r0 = (int64_t)blockIdx.x * GQ;
    const T *Qg = (const T *)a.q + bh * a.Sq * D, *dOg = (const T *)a.d_o + bh * a.Sq * D;
    const T *Kg = (const T *)a.k + bh * a.Skv * D, *Vg = (const T *)a.v + bh * a.Skv * D;
    const float *lse = a.lse_r + bh * a.Sq, *dlt = a.delta + bh * a.Sq;
    const int64_t nrow = MODE == 0 ? a.Sq : a.Skv, ncol = MODE == 0 ? a.Skv : a.Sq;
    const T *Rqg = MODE == 0 ? Qg : Kg, *Rdg = MODE == 0 ? dOg : Vg;
    const T *Cqg = MODE == 0 ? Kg : Qg, *Cdg = MODE == 0 ? Vg : dOg;
    const int t = threadIdx.x;
    for (int i = t; i < GQ * D; i += 256) {
        const int r = i / D, d = i % D;
        const bool ok = r0 + r < nrow;
        Rq[r * DP + d] = ok ? t_load<T>(Rqg + (r0 + r) * D + d) : 0.f;
        Rd[r * DP + d] = ok ? t_load<T>(Rdg + (r0 + r) * D + d) : 0.f;
    }
    const int orow = t / 16, ocol = t % 16;
    float acc0[16], acc1[16]; // MODE 0: dQ | MODE 1: dK, dV
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }

    // column range: MODE 0 keys 0..min(Skv, last query + 1); MODE 1 queries from first key .. Sq
    int64_t c_begin = 0, c_end = ncol;
    if (MODE == 0) {
        const int64_t q_hi = r0 + GQ - 1 < a.Sq - 1 ? r0 + GQ - 1 : a.Sq - 1;
        c_end = a.Skv < q_hi + 1 ? a.Skv : q_hi + 1;
    } else {
        c_begin = r0 / GK * GK;
    }
    for (int64_t c0 = c_begin; c0 < c_end; c0 += GK) {
        __syncthreads();
        for (int i = t; i < GK * D; i += 256) {
            const int r = i / D, d = i % D;
            const bool ok = c0 + r < ncol;
            Cq[r * DP + d] = ok ? t_load<T>(Cqg + (c0 + r) * D + d) : 0.f;
            Cd[r * DP + d] = ok ? t_load<T>(Cdg + (c0 + r) * D + d) : 0.f;
        }
        if (MODE == 1 && t < GK) {
            const bool ok = c0 + t < ncol;
            cl[t] = ok ? lse[c0 + t] : 0.f;
            cd[t] = ok ? dlt[c0 + t] : 0.f;
        }
        __syncthreads();
        {
            const int cc = t % GK;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int rr = t / GK + 8 * h;
                float s = 0.f, dp = 0.f;
                for (int d = 0; d < D; ++d) {
                    s = fmaf(Rq[rr * DP + d], Cq[cc * DP + d], s);
                    dp = fmaf(Rd[rr * DP + d], Cd[cc * DP + d], dp);
                }
                const int64_t mq = MODE == 0 ? r0 + rr : c0 + cc; // query index
                const int64_t nk = MODE == 0 ? c0 + cc : r0 + rr; // key index
                float p = 0.f, ds = 0.f;
                if (nk <= mq && mq < a.Sq && nk < a.Skv) {
                    const float l = MODE == 0 ? lse[mq] : cl[cc];
                    const float dl = MODE == 0 ? dlt[mq] : cd[cc];
                    p = expf(s * a.scale - l);
                    ds = p * (dp - dl);
                }
                Ps[rr * (GK + 1) + cc] = p;
                Ds[rr * (GK + 1) + cc] = ds;
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = ocol + 16 * j;
            if (d < D) {
                float x0 = acc0[j], x1 = acc1[j];
#pragma unroll 4
                for (int cc = 0; cc < GK; ++cc) {
                    // MODE 0: dQ += dS K ; MODE 1: dK += dS^T Q, dV += P^T dO
                    x0 = fmaf(Ds[orow * (GK + 1) + cc], Cq[cc * DP + d], x0);
                    if (MODE == 1) x1 = fmaf(Ps[orow * (GK + 1) + cc], Cd[cc * DP + d], x1);
                }
                acc0[j] = x0;
                acc1[j] = x1;
            }
        }
    }
    const int64_t r = r0 + orow;
    if (r < nrow) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = ocol + 16 * j;
            if (d < D) {
                if (MODE == 0) {
                    t_store<T>((T *)a.dq + (bh * a.Sq + r) * D + d, acc0[j] * a.scale);
                } else {
                    t_store<T>((T *)a.dk + (bh * a.Skv + r) * D + d, acc0[j] * a.scale);
                    t_store<T>((T *)a.dv + (bh * a.Skv + r) * D + d, acc1[j]);
                }
            }
        }
    }
}

static bool mfma_ok(int dtype, int64_t Sq, int64_t Skv, int64_t D) {
    return (dtype == KF_BF16 || dtype == KF_F16) && (D == AD || D == 64) && Sq % 128 == 0 && Skv % 128 == 0 && Sq > 0 && Skv > 0;
}
// Round 6: the generated streams (attn_fwd_w4 / attn_bwd_dkv_w4) and the stored-dS dQ kernel take ANY sequence lengths with Skv >= Sq - rows
// beyond a tensor's end are zero-filled / dropped by the buffer descriptors' range check, and a zero key lies above every real query's
// diagonal when Skv >= Sq (with fewer keys than queries a ragged key count would need a key-length mask: those shapes keep the tiers below).
static bool w4_any_ok(int dtype, int64_t Sq, int64_t Skv, int64_t D) {
    return (dtype == KF_BF16 || dtype == KF_F16) && (D == AD || D == 64) && Sq > 0 && Skv >= Sq;
}

static inline size_t a_align(size_t v) { return (v + 255) / 256 * 256; }

// Backward workspace of the 16-bit matrix-core path: three rows of statistics (O(B H S), always) + as much of dS as the caller gives.
// The dS form (2 products for dQ instead of the 6 a recomputing kernel executes) processes the (batch, head) pairs in GROUPS of as
// many pairs as the workspace holds dS for (multiples of 8 when it can, so a group's heads still pin to XCDs); a workspace that
// cannot hold one pair's dS - or KF_ATTN_SPLIT_BWD - selects the recomputing dQ kernel, whose workspace is the statistics alone.
// kf_attn_bwd_workspace_bytes recommends statistics + min(all of dS, KF_ATTN_DS_CAP_MB (default 16 GiB)): the workspace is bounded
// whatever B, H and S are, and ANY size >= the statistics is accepted (both head sizes, every S).
static inline int64_t stat_rows(int64_t Sq) { return (Sq + 31) / 32 * 32; } // rows per head of the row-constant arrays (AttnArgs::Sqc)
static size_t bwd_stats_bytes(int64_t nbh, int64_t Sq) { // delta [B H, Sq] | -lse / scale [B H, Sqc] | -delta [B H, Sqc]
    return a_align((size_t)nbh * Sq * sizeof(float)) + 2 * a_align((size_t)nbh * stat_rows(Sq) * sizeof(float));
}
// Which layout the dS tiles take. FULL ROWS (the rectangle of rounds 2-5) when the workspace holds them for every pair of the launch:
// the dQ kernel streams 3-12 % faster from rows that start on 2 MiB boundaries (bench.py context, four interleaved runs on two boxes: 0.88 against 0.94-0.99 ms;
// profiles/r06_ab_ds_layout.txt). Else THE CAUSAL HALF (round 6): 0.53 of the bytes at Sq = Skv = 4096 - twice the pairs per group under any cap, and C3 in one
// group of 4.25 GiB where the rectangle needs 8. KF_ATTN_DS_TRI=1 takes the half whatever the workspace holds (kf_attn_bwd_workspace_bytes then asks for it).
// Results do not depend on the layout (bit-identical).
static int ds_tri_mode(int64_t nbh, int64_t Sq, int64_t Skv, size_t budget) { return (knob(KNOB_ATTN_DS_TRI) || budget < ds_bytes(nbh, Sq, Skv, 0)) ? 1 : 0; }
static int64_t ds_group(int64_t nbh, int64_t Sq, int64_t Skv, size_t budget, int tri) { // pairs whose dS fit into `budget` bytes
    const size_t one = ds_bytes(1, Sq, Skv, tri);
    int64_t g = (int64_t)(budget / one);
    if (g >= nbh) return nbh;
    if (g >= 8) g -= g % 8;
    return g;
}
static size_t ds_cap() { // KF_ATTN_DS_CAP_MB, clamped to [0, 4 Ti MB): a negative or absurd value must not wrap the shift
    const long long mb = knob_int(KNOB_ATTN_DS_CAP_MB, 16384);
    return (size_t)std::min<long long>(std::max<long long>(mb, 0), 4ll << 20) << 20;
}

template <typename K>
static int set_lds(K kernel, size_t bytes) { return ensure_dynamic_lds((const void *)kernel, (int)bytes); }

} // namespace kf

using namespace kf;

#ifdef KF_MUTANT
static int g_mutant = 0; // see KF_MUT above
extern "C" int kfmut_select(int which) { g_mutant = which; return KF_OK; }
#endif

static int check_common(const char *who, int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D) {
    KF_REQUIRE(dtype == KF_F32 || dtype == KF_BF16 || dtype == KF_F16, KF_ERR_UNSUPPORTED, "%s: dtype %d not supported", who, dtype);
    KF_REQUIRE(B >= 0 && H >= 0 && Sq >= 0 && Skv >= 0 && D > 0, KF_ERR_INVALID, "%s: bad extents", who);
    KF_REQUIRE(D <= 256, KF_ERR_UNSUPPORTED, "%s: head size %lld > 256", who, (long long)D);
    KF_REQUIRE(B * H <= 65535, KF_ERR_UNSUPPORTED, "%s: B*H %lld > 65535", who, (long long)(B * H));
    return KF_OK;
}

extern "C" int kf_attn_fwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q,
                           const void *k, const void *v, void *o, float *lse, void *stream) {
    return kf_attn_fwd_scaled(dtype, B, H, Sq, Skv, D, D > 0 ? 1.0f / sqrtf((float)D) : 1.0f, q, k, v, o, lse, stream);
}

static AttnArgs::Lay lay_contig(int64_t H, int64_t S, int64_t D, int es) { return {H * S * D * es, S * D * es, D * es}; }
static bool lay_from(const kf_attn_layout *l, int es, AttnArgs::Lay &out) {
    if (!l || l->batch < 0 || l->head < 0 || l->row < 0) return false;
    out = {l->batch * es, l->head * es, l->row * es};
    return out.sb % 16 == 0 && out.sh % 16 == 0 && out.sr % 16 == 0; // 16-byte row pieces
}

static int attn_fwd_impl(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q, const void *k,
                         const void *v, void *o, float *lse, const AttnArgs::Lay *lays, void *stream);

extern "C" int kf_attn_fwd_scaled(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                                  const void *k, const void *v, void *o, float *lse, void *stream) {
    return attn_fwd_impl(dtype, B, H, Sq, Skv, D, scale, q, k, v, o, lse, nullptr, stream);
}

extern "C" int kf_attn_fwd_strided(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                                   const kf_attn_layout *lq, const void *k, const kf_attn_layout *lk, const void *v, const kf_attn_layout *lv,
                                   void *o, const kf_attn_layout *lo, float *lse, void *stream) {
    KF_REQUIRE(mfma_ok(dtype, Sq, Skv, D) || w4_any_ok(dtype, Sq, Skv, D), KF_ERR_UNSUPPORTED,
               "kf_attn_fwd_strided: strided layouts are served by the 16-bit matrix-core kernels only (D = 64 or 128; Skv >= Sq, or Sq, Skv multiples of 128)");
    AttnArgs::Lay lays[4];
    KF_REQUIRE(lay_from(lq, 2, lays[0]) && lay_from(lk, 2, lays[1]) && lay_from(lv, 2, lays[2]) && lay_from(lo, 2, lays[3]), KF_ERR_INVALID,
               "kf_attn_fwd_strided: strides must be non-negative multiples of 8 elements");
    KF_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0, KF_ERR_INVALID, "kf_attn_fwd_strided: operands must be 16-byte aligned");
    return attn_fwd_impl(dtype, B, H, Sq, Skv, D, scale, q, k, v, o, lse, lays, stream);
}

static int attn_fwd_impl(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q, const void *k,
                         const void *v, void *o, float *lse, const AttnArgs::Lay *lays, void *stream) {
    int rc = check_common("kf_attn_fwd", dtype, B, H, Sq, Skv, D);
    if (rc != KF_OK) return rc;
    if (B * H == 0 || Sq == 0) return KF_OK;
    KF_REQUIRE(Skv > 0, KF_ERR_INVALID, "kf_attn_fwd: Skv must be positive");
    KF_REQUIRE(q && k && v && o, KF_ERR_INVALID, "kf_attn_fwd: null operand");
    hipStream_t st = as_stream(stream);
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const char *)q; a.k = (const char *)k; a.v = (const char *)v; a.out = (char *)o; a.lse = lse;
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.D = D;
    a.nbh = (int)(B * H);
    KF_REQUIRE(scale > 0.f && scale < INFINITY, KF_ERR_INVALID, "attention: the softmax scale must be positive and finite");
    a.scale = scale;
    a.scale_log2e = scale * kLog2e;
#ifdef KF_FWD_W4_STAMPS
    extern unsigned long long *kf_attn_tl_host_ptr();
    a.dbg = kf_attn_tl_host_ptr();
#endif
    a.xcd_map = ((B * H) % 8 == 0) && !knob(KNOB_ATTN_NO_XCD);
    a.defer = knob(KNOB_ATTN_NO_DEFER) ? -INFINITY : kDeferMax; // A/B switch: rescale O at every tile
#ifdef KF_MUTANT
    a.mutant = g_mutant;
#endif
    if (lays) { a.lq = lays[0]; a.lk = lays[1]; a.lv = lays[2]; a.lo = lays[3]; }
    else { a.lq = a.lo = lay_contig(H, Sq, D, 2); a.lk = a.lv = lay_contig(H, Skv, D, 2); }
    // the generated stream's shape conditions (KF_ATTN_FWD_V3 keeps the 8-wave kernel where that one can run: A/B)
    const bool fwd_w4 = w4_any_ok(dtype, Sq, Skv, D) && a.lk.sr == a.lv.sr && (uint64_t)(Skv + FQ) * (uint64_t)a.lk.sr < (1ull << 32) &&
                        (uint64_t)FQ * (uint64_t)std::max(a.lq.sr, a.lo.sr) < (1ull << 31) && !knob(KNOB_ATTN_FWD_V3);
    if (mfma_ok(dtype, Sq, Skv, D) || fwd_w4) {
        const size_t lds3 = SRING * FBUF;
        const int64_t nxb3 = (Sq + FQ - 1) / FQ;
        a.persist = (nxb3 % 2 == 0 && nxb3 >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
        dim3 grid3((unsigned)((a.persist ? nxb3 / (2 * a.persist) : nxb3) * B * H));
        a.nvwg = grid3.x;
        dim3 grid3w = grid3; // the generated kernels' real grid: all virtual workgroups, or KF_ATTN_GRID_WGS of them looping (a multiple of 8)
        if (const long gw = knob_int(KNOB_ATTN_GRID_WGS, 0); gw >= 8 && (unsigned)gw < grid3.x) grid3w.x = (unsigned)(gw / 8 * 8);
        KF_PROF(D == 64 ? "attn_fwd_mfma_d64" : "attn_fwd_mfma", st);
        // round 4: the one-wave-per-SIMD stream (attn_fwd_w4_kernel) wherever its shape conditions hold; KF_ATTN_FWD_V3 keeps the 8-wave kernel (A/B)
        if (fwd_w4) {
#define KF_FWD4(BF_, SQ_)                                                                                   \
    {                                                                                                       \
        if ((rc = set_lds(attn_fwd_w4_kernel<BF_, SQ_>, KF_FWD_W4_LDS_BYTES)) != KF_OK) return rc;           \
        attn_fwd_w4_kernel<BF_, SQ_><<<grid3w, 256, KF_FWD_W4_LDS_BYTES, st>>>(a);                           \
    }
            const bool sq = D == AD && knob(KNOB_ATTN_SCALED_OPERANDS); // opt-in: c q rounded once per pass (faster; score error grows with the logits)
            if (D == 64) {
                if (dtype == KF_BF16) {
                    if ((rc = set_lds(attn_fwd_w4_kernel<true, false, true>, KF_FWD_W4_LDS_BYTES)) != KF_OK) return rc;
                    attn_fwd_w4_kernel<true, false, true><<<grid3w, 256, KF_FWD_W4_LDS_BYTES, st>>>(a);
                } else {
                    if ((rc = set_lds(attn_fwd_w4_kernel<false, false, true>, KF_FWD_W4_LDS_BYTES)) != KF_OK) return rc;
                    attn_fwd_w4_kernel<false, false, true><<<grid3w, 256, KF_FWD_W4_LDS_BYTES, st>>>(a);
                }
            } else if (dtype == KF_BF16) { if (sq) KF_FWD4(true, true) else KF_FWD4(true, false) }
            else { if (sq) KF_FWD4(false, true) else KF_FWD4(false, false) }
#undef KF_FWD4
            KF_LAUNCH_CHECK();
            return KF_OK;
        }
#define KF_FWD(BF_, D_)                                                                  \
    {                                                                                    \
        if ((rc = set_lds(attn_fwd_v3_kernel<BF_, D_>, lds3)) != KF_OK) return rc;       \
        attn_fwd_v3_kernel<BF_, D_><<<grid3, FNT, lds3, st>>>(a);                         \
    }
        if (dtype == KF_BF16) { if (D == 64) KF_FWD(true, 64) else KF_FWD(true, 128) }
        else { if (D == 64) KF_FWD(false, 64) else KF_FWD(false, 128) }
#undef KF_FWD
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    // (from here on: kernels of contiguous [B, H, S, D] tensors only)
    KF_REQUIRE(!lays, KF_ERR_UNSUPPORTED, "kf_attn_fwd_strided: this shape / stride combination has no matrix-core kernel (ragged lengths want K and V with one row stride)");
    if (dtype == KF_F32 && (D == 64 || D == 128) && Sq % 32 == 0 && Skv % 32 == 0 && !knob(KNOB_ATTN_F32_GENERIC)) {
        // the reference's own fast path (f32, head size 64 or 128): exact-f32 MFMA
        const size_t ldsx = std::max((size_t)2 * XK * (D + 4), (size_t)4 * 32 * (D + 4)) * sizeof(float);
        const int64_t nxx = (Sq + XQ - 1) / XQ;
        a.persist = (nxx % 2 == 0 && nxx >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
        dim3 gridx((unsigned)((a.persist ? nxx / 2 : nxx) * B * H));
        KF_PROF("attn_fwd_f32_mfma", st);
        if (D == 128) {
            if ((rc = set_lds(attn_fwd_f32_mfma_kernel<128>, ldsx)) != KF_OK) return rc;
            attn_fwd_f32_mfma_kernel<128><<<gridx, 256, ldsx, st>>>(a);
        } else {
            if ((rc = set_lds(attn_fwd_f32_mfma_kernel<64>, ldsx)) != KF_OK) return rc;
            attn_fwd_f32_mfma_kernel<64><<<gridx, 256, ldsx, st>>>(a);
        }
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    const int DP = (int)D + 1;
    const size_t lds = sizeof(float) * ((size_t)GQ * DP + (size_t)GK * DP + (size_t)GK * D + (size_t)GQ * (GK + 1) + GQ);
    dim3 grid((unsigned)((Sq + GQ - 1) / GQ), (unsigned)(B * H));
    KF_PROF("attn_fwd_generic", st);
    switch (dtype) {
    case KF_F32:
        rc = set_lds(attn_fwd_generic_kernel<float>, lds);
        if (rc != KF_OK) return rc;
        attn_fwd_generic_kernel<float><<<grid, 256, lds, st>>>(a);
        break;
    case KF_BF16:
        rc = set_lds(attn_fwd_generic_kernel<bf16_t>, lds);
        if (rc != KF_OK) return rc;
        attn_fwd_generic_kernel<bf16_t><<<grid, 256, lds, st>>>(a);
        break;
    default:
        rc = set_lds(attn_fwd_generic_kernel<f16_t>, lds);
        if (rc != KF_OK) return rc;
        attn_fwd_generic_kernel<f16_t><<<grid, 256, lds, st>>>(a);
        break;
    }
    KF_LAUNCH_CHECK();
    return KF_OK;
}

extern "C" int kf_attn_bwd_workspace_bytes(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D,
                                           size_t *bytes) {
    KF_REQUIRE(bytes, KF_ERR_INVALID, "kf_attn_bwd_workspace_bytes: null out pointer");
    int rc = check_common("kf_attn_bwd_workspace_bytes", dtype, B, H, Sq, Skv, D);
    if (rc != KF_OK) return rc;
    *bytes = bwd_stats_bytes(B * H, Sq); // delta | -lse log2(e) | -delta
    if ((mfma_ok(dtype, Sq, Skv, D) || w4_any_ok(dtype, Sq, Skv, D)) && !knob(KNOB_ATTN_SPLIT_BWD) && B * H > 0)
    { // + dS in 16 bits (DS_TILE): full rows for every pair when the cap allows (the faster dQ stream), else the causal half of as many pairs as the cap holds
        const int tri = ds_tri_mode(B * H, Sq, Skv, ds_cap());
        *bytes += ds_bytes(ds_group(B * H, Sq, Skv, ds_cap(), tri), Sq, Skv, tri);
    }
    return KF_OK;
}

extern "C" int kf_attn_bwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q,
                           const void *k, const void *v, const void *o, const float *lse, const void *d_o, void *dq,
                           void *dk, void *dv, void *workspace, size_t workspace_bytes, void *stream) {
    return kf_attn_bwd_scaled(dtype, B, H, Sq, Skv, D, D > 0 ? 1.0f / sqrtf((float)D) : 1.0f, q, k, v, o, lse, d_o, dq, dk, dv, workspace,
                              workspace_bytes, stream);
}

static int attn_bwd_impl(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q, const void *k,
                         const void *v, const void *o, const float *lse, const void *d_o, void *dq, void *dk, void *dv, const AttnArgs::Lay *lays,
                         void *workspace, size_t workspace_bytes, void *stream);

extern "C" int kf_attn_bwd_scaled(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                                  const void *k, const void *v, const void *o, const float *lse, const void *d_o, void *dq,
                                  void *dk, void *dv, void *workspace, size_t workspace_bytes, void *stream) {
    return attn_bwd_impl(dtype, B, H, Sq, Skv, D, scale, q, k, v, o, lse, d_o, dq, dk, dv, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int kf_attn_bwd_strided(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                                   const kf_attn_layout *lq, const void *k, const kf_attn_layout *lk, const void *v, const kf_attn_layout *lv,
                                   const void *o, const kf_attn_layout *lo, const float *lse, const void *d_o, const kf_attn_layout *ldo, void *dq,
                                   const kf_attn_layout *ldq, void *dk, const kf_attn_layout *ldk, void *dv, const kf_attn_layout *ldv,
                                   void *workspace, size_t workspace_bytes, void *stream) {
    KF_REQUIRE(mfma_ok(dtype, Sq, Skv, D) || w4_any_ok(dtype, Sq, Skv, D), KF_ERR_UNSUPPORTED,
               "kf_attn_bwd_strided: strided layouts are served by the 16-bit matrix-core kernels only (D = 64 or 128; Skv >= Sq, or Sq, Skv multiples of 128)");
    AttnArgs::Lay lays[8];
    const kf_attn_layout *in[8] = {lq, lk, lv, lo, ldo, ldq, ldk, ldv};
    for (int i = 0; i < 8; ++i)
        KF_REQUIRE(lay_from(in[i], 2, lays[i]), KF_ERR_INVALID, "kf_attn_bwd_strided: strides must be non-negative multiples of 8 elements");
    KF_REQUIRE(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0,
               KF_ERR_INVALID, "kf_attn_bwd_strided: operands must be 16-byte aligned");
    return attn_bwd_impl(dtype, B, H, Sq, Skv, D, scale, q, k, v, o, lse, d_o, dq, dk, dv, lays, workspace, workspace_bytes, stream);
}

static int attn_bwd_impl(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q, const void *k,
                         const void *v, const void *o, const float *lse, const void *d_o, void *dq, void *dk, void *dv, const AttnArgs::Lay *lays,
                         void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_common("kf_attn_bwd", dtype, B, H, Sq, Skv, D);
    if (rc != KF_OK) return rc;
    if (B * H == 0 || Sq == 0 || Skv == 0) return KF_OK;
    KF_REQUIRE(q && k && v && o && lse && d_o && dq && dk && dv, KF_ERR_INVALID, "kf_attn_bwd: null operand");
    const size_t need = bwd_stats_bytes(B * H, Sq); // the minimum; what lies beyond it holds dS (see ds_group)
    KF_REQUIRE(workspace && workspace_bytes >= need, KF_ERR_WORKSPACE, "kf_attn_bwd: workspace of at least %zu bytes required, got %zu", need, workspace_bytes);
    hipStream_t st = as_stream(stream);
    AttnArgs a;
    memset(&a, 0, sizeof(a));
    a.q = (const char *)q; a.k = (const char *)k; a.v = (const char *)v; a.o = (const char *)o; a.d_o = (const char *)d_o;
    a.dq = (char *)dq; a.dk = (char *)dk; a.dv = (char *)dv;
    a.lse_r = lse; a.delta = (float *)workspace;
    a.Sqc = stat_rows(Sq);
    a.nlse = (float *)((char *)workspace + a_align((size_t)B * H * Sq * sizeof(float)));
    a.ndelta = (float *)((char *)a.nlse + a_align((size_t)B * H * a.Sqc * sizeof(float)));
    a.xcd_map = ((B * H) % 8 == 0) && !knob(KNOB_ATTN_NO_XCD);
    a.B = B; a.H = H; a.Sq = Sq; a.Skv = Skv; a.D = D;
    a.nbh = (int)(B * H);
    KF_REQUIRE(scale > 0.f && scale < INFINITY, KF_ERR_INVALID, "attention: the softmax scale must be positive and finite");
    a.scale = scale;
    a.scale_log2e = scale * kLog2e;
#ifdef KF_DKV_W4_STAMPS
    extern unsigned long long *kf_attn_tl_host_ptr();
    a.dbg = kf_attn_tl_host_ptr();
#endif
#ifdef KF_MUTANT
    a.mutant = g_mutant;
#endif
    if (lays) { a.lq = lays[0]; a.lk = lays[1]; a.lv = lays[2]; a.lo = lays[3]; a.ldo = lays[4]; a.ldq = lays[5]; a.ldk = lays[6]; a.ldv = lays[7]; }
    else { a.lq = a.lo = a.ldo = a.ldq = lay_contig(H, Sq, D, 2); a.lk = a.lv = a.ldk = a.ldv = lay_contig(H, Skv, D, 2); }
    const int64_t nrows = B * H * Sq;
    const bool tiled = mfma_ok(dtype, Sq, Skv, D); // whole 128-row tiles: every 16-bit matrix-core kernel can run
    // round 4: 64 keys per wave (attn_bwd_dkv_w4_kernel) wherever its shape conditions hold; KF_ATTN_DKV_V4 keeps the 32-key kernel (A/B).
    // round 6: any Sq, Skv with Skv >= Sq (w4_any_ok: rows beyond a tensor's end are zero-filled / dropped by the descriptors)
    const bool dkv_w4 = ((tiled && Skv % K5B == 0) || w4_any_ok(dtype, Sq, Skv, D)) && a.lk.sr == a.lv.sr && a.ldk.sr == a.ldv.sr &&
                        (uint64_t)(Sq + 32) * (uint64_t)std::max(a.lq.sr, a.ldo.sr) < (1ull << 32) && (uint64_t)K5B * (uint64_t)std::max(a.lk.sr, a.ldk.sr) < (1ull << 31) &&
                        (uint64_t)((const char *)a.ndelta - (const char *)a.nlse) < (1ull << 31) && !knob(KNOB_ATTN_DKV_V4);
    // a ragged shape has no other matrix-core kernels: it needs the generated dK/dV stream AND room for dS (the recomputing dQ kernel wants whole tiles)
    const int ds_tri = ds_tri_mode(B * H, Sq, Skv, workspace_bytes - need);
    const bool ragged_ok = !tiled && dkv_w4 && !knob(KNOB_ATTN_SPLIT_BWD) && ds_group(B * H, Sq, Skv, workspace_bytes - need, ds_tri) > 0;
    if (tiled || ragged_ok) {
        const bool bf = dtype == KF_BF16;
        // exact f32 scores everywhere by default (exponent = (s - lse / scale) * scale log2 e: the row constant is -lse / scale); only the opt-in
        // scaled-K form of the generated stream scales K by scale log2 e once per block and wants -lse log2 e
        const bool dkv_sq = dkv_w4 && D == AD && knob(KNOB_ATTN_SCALED_OPERANDS);
        const float rscale = dkv_sq ? kLog2e : 1.0f / scale;
        {
            KF_PROF("attn_bwd_delta", st);
            // two rows per 16-lane group (four 16-byte loads in flight per lane): 0.110 -> 0.101 ms at C3; four rows: 0.100
            const int64_t nrows_c = B * H * a.Sqc;   // (rows of the padded row-constant arrays: the pad rows are written as zeros)
            const unsigned gd2 = (unsigned)((nrows_c + 31) / 32);
            if (bf) attn_delta_kernel<true, 2><<<gd2, 256, 0, st>>>(a.o, a.d_o, a.delta, nrows_c, a.lse_r, a.nlse, a.ndelta, rscale, a.lo, a.ldo, Sq, H, (int)(D / 8), a.Sqc);
            else attn_delta_kernel<false, 2><<<gd2, 256, 0, st>>>(a.o, a.d_o, a.delta, nrows_c, a.lse_r, a.nlse, a.ndelta, rscale, a.lo, a.ldo, Sq, H, (int)(D / 8), a.Sqc);
            KF_LAUNCH_CHECK();
        }
        const int64_t nbh = B * H;
        const int64_t group = knob(KNOB_ATTN_SPLIT_BWD) ? 0 : ds_group(nbh, Sq, Skv, workspace_bytes - need, ds_tri);
        const bool keep_ds = group > 0;
        a.ds = keep_ds ? (char *)workspace + need : nullptr;
        a.ds_tri = ds_tri;
        a.ds_pair = ds_pair_tiles(Sq, Skv, ds_tri);
        a.ds_nkb = (Skv + 255) / 256;
        const int64_t nkb4 = Skv / K4B, nxq = (Sq + FQ - 1) / FQ;
        const int pair_kv = (nkb4 % 2 == 0 && nkb4 >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
        const int pair_q = (nxq % 2 == 0 && nxq >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
#define KF_DKV(BF_, DS_, D_)                                                                  \
    {                                                                                         \
        if ((rc = set_lds(attn_bwd_dkv_v4_kernel<BF_, DS_, D_>, K4LDS)) != KF_OK) return rc;  \
        attn_bwd_dkv_v4_kernel<BF_, DS_, D_><<<gk4, 256, K4LDS, st>>>(a);                     \
    }
#define KF_DQ(BF_, D_)                                                                      \
    {                                                                                       \
        if ((rc = set_lds(attn_bwd_dq_ds_kernel<BF_, D_>, DQ_LDS)) != KF_OK) return rc;     \
        attn_bwd_dq_ds_kernel<BF_, D_><<<gq2, FNT, DQ_LDS, st>>>(a);                         \
    }
#define KF_DQ2(BF_, D_)                                                                     \
    {                                                                                       \
        if ((rc = set_lds(attn_bwd_dq_v2_kernel<BF_, D_>, QLDS)) != KF_OK) return rc;       \
        attn_bwd_dq_v2_kernel<BF_, D_><<<gq2, FNT, QLDS, st>>>(a);                           \
    }
        // (Tried in round 3 and removed: the dS workspace as two half-group slots with group g's dQ - HBM-bound, it streams dS - on a second
        // stream beside group g + 1's matrix-bound dK/dV. At C3 the backward took 3.39-3.78 ms against 3.32-3.36 in sequence
        // (profiles/r03_attn_bwd_overlap_experiment.txt): both kernels fill every CU, so the dispatcher interleaves them instead of
        // running them side by side, and each group boundary adds a tail.)
        const int64_t grp = keep_ds ? group : nbh;
        // one group of (batch, head) pairs at a time: dK/dV (stores the group's dS), then dQ from it. Without dS: one group, all pairs.
        for (int64_t bh0 = 0; bh0 < nbh; bh0 += grp) {
            a.bh0 = bh0;
            a.nbh = (int)std::min<int64_t>(grp, nbh - bh0);
            a.xcd_map = (a.nbh % 8 == 0) && !knob(KNOB_ATTN_NO_XCD);
            { // dK / dV: one wave per SIMD, pinned MFMA / VALU interleave
                a.persist = pair_kv;
                a.persist_rev = 1; // the short block of the pair first: 2.17 ms against 2.32 the other way round (2.22 unpaired)
                dim3 gk4((unsigned)((a.persist ? nkb4 / 2 : nkb4) * a.nbh));
                KF_PROF(D == 64 ? "attn_bwd_dkv_mfma_d64" : "attn_bwd_dkv_mfma", st);
                const int64_t nkb5 = (Skv + K5B - 1) / K5B;
                if (dkv_w4) {
                    a.persist = (nkb5 % 2 == 0 && nkb5 >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
                    dim3 gk5((unsigned)((a.persist ? nkb5 / 2 : nkb5) * a.nbh));
                    a.nvwg = gk5.x;
                    if (const long gw = knob_int(KNOB_ATTN_GRID_WGS, 0); gw >= 8 && (unsigned)gw < gk5.x) gk5.x = (unsigned)(gw / 8 * 8);
#define KF_DKV5(BF_, DS_, SQ_)                                                                                    \
    {                                                                                                             \
        if ((rc = set_lds(attn_bwd_dkv_w4_kernel<BF_, DS_, SQ_>, KF_DKV_W4_LDS_BYTES)) != KF_OK) return rc;        \
        attn_bwd_dkv_w4_kernel<BF_, DS_, SQ_><<<gk5, 256, KF_DKV_W4_LDS_BYTES, st>>>(a);                          \
    }
#define KF_DKV5D(BF_, DS_)                                                                                              \
    {                                                                                                                   \
        if ((rc = set_lds(attn_bwd_dkv_w4_kernel<BF_, DS_, false, true>, KF_DKV_W4_LDS_BYTES)) != KF_OK) return rc;     \
        attn_bwd_dkv_w4_kernel<BF_, DS_, false, true><<<gk5, 256, KF_DKV_W4_LDS_BYTES, st>>>(a);                        \
    }
#define KF_DKV5S(BF_, DS_) { if (D == 64) KF_DKV5D(BF_, DS_) else if (dkv_sq) KF_DKV5(BF_, DS_, true) else KF_DKV5(BF_, DS_, false) }
                    if (bf) { if (keep_ds) KF_DKV5S(true, true) else KF_DKV5S(true, false) }
                    else { if (keep_ds) KF_DKV5S(false, true) else KF_DKV5S(false, false) }
#undef KF_DKV5S
#undef KF_DKV5D
#undef KF_DKV5
                } else if (D == 64) {
                    if (bf) { if (keep_ds) KF_DKV(true, true, 64) else KF_DKV(true, false, 64) }
                    else { if (keep_ds) KF_DKV(false, true, 64) else KF_DKV(false, false, 64) }
                } else if (bf) { if (keep_ds) KF_DKV(true, true, 128) else KF_DKV(true, false, 128) }
                else { if (keep_ds) KF_DKV(false, true, 128) else KF_DKV(false, false, 128) }
                KF_LAUNCH_CHECK();
            }
            a.persist = pair_q;
            a.persist_rev = 0;
            dim3 gq2((unsigned)((a.persist ? nxq / 2 : nxq) * a.nbh));
            if (keep_ds) { // dQ = scale dS K from the stored dS
                KF_PROF(D == 64 ? "attn_bwd_dq_mfma_d64" : "attn_bwd_dq_mfma", st);
                if (bf) { if (D == 64) KF_DQ(true, 64) else KF_DQ(true, 128) }
                else { if (D == 64) KF_DQ(false, 64) else KF_DQ(false, 128) }
                KF_LAUNCH_CHECK();
            } else { // the recomputing dQ kernel (S and dP again)
                KF_PROF(D == 64 ? "attn_bwd_dq_mfma_split_d64" : "attn_bwd_dq_mfma_split", st);
                if (bf) { if (D == 64) KF_DQ2(true, 64) else KF_DQ2(true, 128) }
                else { if (D == 64) KF_DQ2(false, 64) else KF_DQ2(false, 128) }
                KF_LAUNCH_CHECK();
            }
        }
#undef KF_DKV
#undef KF_DQ
#undef KF_DQ2
        return KF_OK;
    }
    // (from here on: kernels of contiguous [B, H, S, D] tensors only)
    KF_REQUIRE(!lays, KF_ERR_UNSUPPORTED, "kf_attn_bwd_strided: this shape / stride / workspace combination has no matrix-core kernel (ragged lengths want room for dS)");
    const unsigned gd = (unsigned)((nrows + 3) / 4);
    if (dtype == KF_F32 && (D == 64 || D == 128) && Sq % 32 == 0 && Skv % 32 == 0 && !knob(KNOB_ATTN_F32_GENERIC)) {
        // exact-f32 MFMA backward (the f32 forward's counterpart; the reference has no backward)
        {
            KF_PROF("attn_bwd_delta", st);
            attn_delta_generic_kernel<float><<<gd, 256, 0, st>>>((const float *)o, (const float *)d_o, a.delta, nrows, (int)D);
            KF_LAUNCH_CHECK();
        }
        const size_t tiles = (size_t)2 * XK * (D + 4) * sizeof(float) + 256, slabs = (size_t)4 * 32 * (D + 4) * sizeof(float);
        const size_t ldsx = std::max(tiles, slabs);
        const int64_t nxq = (Sq + XQ - 1) / XQ, nkb = (Skv + XQ - 1) / XQ;
        {
            a.persist = (nkb % 2 == 0 && nkb >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
            dim3 g((unsigned)((a.persist ? nkb / 2 : nkb) * B * H));
            KF_PROF("attn_bwd_dkv_f32_mfma", st);
            if (D == 128) {
                if ((rc = set_lds(attn_bwd_dkv_f32_mfma_kernel<128>, ldsx)) != KF_OK) return rc;
                attn_bwd_dkv_f32_mfma_kernel<128><<<g, 256, ldsx, st>>>(a);
            } else {
                if ((rc = set_lds(attn_bwd_dkv_f32_mfma_kernel<64>, ldsx)) != KF_OK) return rc;
                attn_bwd_dkv_f32_mfma_kernel<64><<<g, 256, ldsx, st>>>(a);
            }
            KF_LAUNCH_CHECK();
        }
        {
            a.persist = (nxq % 2 == 0 && nxq >= 4 && !knob(KNOB_ATTN_NO_PAIR)) ? 1 : 0;
            dim3 g((unsigned)((a.persist ? nxq / 2 : nxq) * B * H));
            KF_PROF("attn_bwd_dq_f32_mfma", st);
            if (D == 128) {
                if ((rc = set_lds(attn_bwd_dq_f32_mfma_kernel<128>, ldsx)) != KF_OK) return rc;
                attn_bwd_dq_f32_mfma_kernel<128><<<g, 256, ldsx, st>>>(a);
            } else {
                if ((rc = set_lds(attn_bwd_dq_f32_mfma_kernel<64>, ldsx)) != KF_OK) return rc;
                attn_bwd_dq_f32_mfma_kernel<64><<<g, 256, ldsx, st>>>(a);
            }
            KF_LAUNCH_CHECK();
        }
        return KF_OK;
    }
    const int DP = (int)D + 1;
    const size_t lds = sizeof(float) * ((size_t)2 * GQ * DP + (size_t)2 * GK * DP + (size_t)2 * GQ * (GK + 1) + 2 * GK);
    dim3 gq((unsigned)((Sq + GQ - 1) / GQ), (unsigned)(B * H)), gk((unsigned)((Skv + GQ - 1) / GQ), (unsigned)(B * H));
    KF_PROF("attn_bwd_generic", st);
#define KF_GENERIC_BWD(T)                                                                                   \
    attn_delta_generic_kernel<T><<<gd, 256, 0, st>>>((const T *)o, (const T *)d_o, a.delta, nrows, (int)D);  \
    KF_LAUNCH_CHECK();                                                                                      \
    if ((rc = set_lds(attn_bwd_generic_kernel<T, 0>, lds)) != KF_OK) return rc;                             \
    if ((rc = set_lds(attn_bwd_generic_kernel<T, 1>, lds)) != KF_OK) return rc;                             \
    attn_bwd_generic_kernel<T, 0><<<gq, 256, lds, st>>>(a);                                                 \
    KF_LAUNCH_CHECK();                                                                                      \
    attn_bwd_generic_kernel<T, 1><<<gk, 256, lds, st>>>(a);                                                 \
    KF_LAUNCH_CHECK();
    switch (dtype) {
    case KF_F32: { KF_GENERIC_BWD(float) } break;
    case KF_BF16: { KF_GENERIC_BWD(bf16_t) } break;
    default: { KF_GENERIC_BWD(f16_t) } break;
    }
#undef KF_GENERIC_BWD
    return KF_OK;
}

#ifdef KF_ATTN_TIMELINE
static unsigned long long *g_attn_tl_host = nullptr;
unsigned long long *kf_attn_tl_host_ptr() { return g_attn_tl_host; }
extern "C" int kfdbg_attn_timeline(void *buf) { // buf: grid x 2 passes x 8 waves x 8 counters of 8 bytes, zero-filled by the caller
    unsigned long long *p = (unsigned long long *)buf;
    g_attn_tl_host = p;
    return hipMemcpyToSymbol(HIP_SYMBOL(kf::g_attn_tl), &p, sizeof(p)) == hipSuccess ? KF_OK : KF_ERR_HIP;
}
#endif
