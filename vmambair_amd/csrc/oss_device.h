// oss_device.h -- device-side building blocks of the gfx950 selective-scan kernels.
//
// Written for CDNA4 only: 64-lane wavefronts, DPP row/wave controls of the gfx9 ISA family
// (row_shr, row_bcast:15/31, wave_shr/shl), v_exp_f32 / v_log_f32 transcendentals, LDS in 16-byte
// lane-linear images.  No CUB/hipCUB, no compatibility macros.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace oss {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kScanChunk = 256;  // time steps between saved states in x (oss_scan_chunk())
constexpr int kNB = 16;          // states per LDS tile
// lane states (include/vmambair_oss.h: hs): entries per (batch, row, state) line, one per 8 scan steps, padded to whole 512-step chunks
__host__ __device__ inline size_t lane_state_stride(int seqlen) { return (size_t)(((seqlen + 7) / 8 + 63) & ~63); }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct bf16_t { uint16_t v; };
struct f16_t { uint16_t v; };

// ---------------------------------------------------------------------------------------------
// element conversion (reference: Converter::to_float, selective_scan_common.h:56-86 -- all math
// in fp32 whatever the I/O type)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }
__device__ __forceinline__ float to_f32(f16_t x) {
    _Float16 h;
    __builtin_memcpy(&h, &x.v, 2);
    return (float)h;
}
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float x) {
    // v_cvt_pk_bf16_f32 (gfx950): round-to-nearest-even, NaN stays NaN -- torch's float->bfloat16 rounding
    const __bf16 h = (__bf16)x;
    bf16_t r;
    __builtin_memcpy(&r.v, &h, 2);
    return r;
}
template <> __device__ __forceinline__ f16_t from_f32<f16_t>(float x) {
    _Float16 h = (_Float16)x;  // v_cvt_f16_f32, RNE
    f16_t r;
    __builtin_memcpy(&r.v, &h, 2);
    return r;
}

// unpack a 32-bit word holding two 16-bit elements (little endian: element 0 in the low half)
template <typename T> __device__ __forceinline__ void unpack2(uint32_t w, float &lo, float &hi);
template <> __device__ __forceinline__ void unpack2<bf16_t>(uint32_t w, float &lo, float &hi) {
    lo = __uint_as_float(w << 16);
    hi = __uint_as_float(w & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack2<f16_t>(uint32_t w, float &lo, float &hi) {
    lo = to_f32(f16_t{(uint16_t)(w & 0xffffu)});
    hi = to_f32(f16_t{(uint16_t)(w >> 16)});
}
template <typename T> __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    return (uint32_t)from_f32<T>(lo).v | ((uint32_t)from_f32<T>(hi).v << 16);
}
template <> __device__ __forceinline__ uint32_t pack2<bf16_t>(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    const bf16x2 h = __builtin_convertvector(f32x2{lo, hi}, bf16x2);  // one v_cvt_pk_bf16_f32
    uint32_t u;
    __builtin_memcpy(&u, &h, 4);
    return u;
}

// V consecutive elements per lane as one access (V = 2: 4-byte accesses for the 16-bit types -- a wave's 2-byte
// accesses move 128 bytes per instruction and were what bounded these kernels)
template <typename T, int V> __device__ __forceinline__ void load_v(const T *p, float (&v)[V]) {
    static_assert(V == 1 || V == 2 || V == 8, "load_v: 1, 2 or 8 elements");
    if constexpr (V == 1) {
        v[0] = to_f32(*p);
    } else if constexpr (V == 8 && sizeof(T) == 4) {
        const f32x4 a = *reinterpret_cast<const f32x4 *>(p), b = *reinterpret_cast<const f32x4 *>(p + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else if constexpr (V == 8) {   // one 16-byte access
        const u32x4 q = *reinterpret_cast<const u32x4 *>(p);
        unpack2<T>(q.x, v[0], v[1]); unpack2<T>(q.y, v[2], v[3]); unpack2<T>(q.z, v[4], v[5]); unpack2<T>(q.w, v[6], v[7]);
    } else if constexpr (sizeof(T) == 4) {
        const f32x2 q = *reinterpret_cast<const f32x2 *>(p);
        v[0] = q.x; v[1] = q.y;
    } else {
        unpack2<T>(*reinterpret_cast<const uint32_t *>(p), v[0], v[1]);
    }
}
template <typename T, int V> __device__ __forceinline__ void store_v(T *p, const float (&v)[V]) {
    if constexpr (V == 1) {
        *p = from_f32<T>(v[0]);
    } else if constexpr (V == 8 && sizeof(T) == 4) {
        *reinterpret_cast<f32x4 *>(p) = f32x4{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4 *>(p + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else if constexpr (V == 8) {
        *reinterpret_cast<u32x4 *>(p) = u32x4{pack2<T>(v[0], v[1]), pack2<T>(v[2], v[3]), pack2<T>(v[4], v[5]), pack2<T>(v[6], v[7])};
    } else if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<f32x2 *>(p) = f32x2{v[0], v[1]};
    } else {
        *reinterpret_cast<uint32_t *>(p) = pack2<T>(v[0], v[1]);
    }
}

// ---------------------------------------------------------------------------------------------
// I contiguous elements per lane <-> float[I].  `vec` = pointer is 16-byte aligned and all I
// elements are in range; otherwise element-wise with zero fill beyond `valid` elements.
// ---------------------------------------------------------------------------------------------
template <int I>
__device__ __forceinline__ void load_items(const float *p, int valid, bool vec, float (&v)[I]) {
    if (vec) {
#pragma unroll
        for (int k = 0; k < I / 4; ++k) {
            f32x4 q = *reinterpret_cast<const f32x4 *>(p + 4 * k);
            v[4 * k] = q.x; v[4 * k + 1] = q.y; v[4 * k + 2] = q.z; v[4 * k + 3] = q.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i) v[i] = (i < valid) ? p[i] : 0.f;
    }
}
template <int I, typename T>
__device__ __forceinline__ void load_items(const T *p, int valid, bool vec, float (&v)[I]) {
    static_assert(I % 4 == 0, "");
    if (vec) {
        if constexpr (I % 8 == 0) {
#pragma unroll
            for (int k = 0; k < I / 8; ++k) {
                u32x4 q = *reinterpret_cast<const u32x4 *>(p + 8 * k);
                unpack2<T>(q.x, v[8 * k], v[8 * k + 1]);
                unpack2<T>(q.y, v[8 * k + 2], v[8 * k + 3]);
                unpack2<T>(q.z, v[8 * k + 4], v[8 * k + 5]);
                unpack2<T>(q.w, v[8 * k + 6], v[8 * k + 7]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < I / 4; ++k) {
                u32x2 q = *reinterpret_cast<const u32x2 *>(p + 4 * k);
                unpack2<T>(q.x, v[4 * k], v[4 * k + 1]);
                unpack2<T>(q.y, v[4 * k + 2], v[4 * k + 3]);
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i) v[i] = (i < valid) ? to_f32(p[i]) : 0.f;
    }
}
template <int I>
__device__ __forceinline__ void store_items(float *p, int valid, bool vec, const float (&v)[I]) {
    if (vec) {
#pragma unroll
        for (int k = 0; k < I / 4; ++k) {
            f32x4 q = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
            *reinterpret_cast<f32x4 *>(p + 4 * k) = q;
        }
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i)
            if (i < valid) p[i] = v[i];
    }
}
template <int I, typename T>
__device__ __forceinline__ void store_items(T *p, int valid, bool vec, const float (&v)[I]) {
    if (vec) {
        if constexpr (I % 8 == 0) {
#pragma unroll
            for (int k = 0; k < I / 8; ++k) {
                u32x4 q = {pack2<T>(v[8 * k], v[8 * k + 1]), pack2<T>(v[8 * k + 2], v[8 * k + 3]),
                           pack2<T>(v[8 * k + 4], v[8 * k + 5]), pack2<T>(v[8 * k + 6], v[8 * k + 7])};
                *reinterpret_cast<u32x4 *>(p + 8 * k) = q;
            }
        } else {
#pragma unroll
            for (int k = 0; k < I / 4; ++k) {
                u32x2 q = {pack2<T>(v[4 * k], v[4 * k + 1]), pack2<T>(v[4 * k + 2], v[4 * k + 3])};
                *reinterpret_cast<u32x2 *>(p + 4 * k) = q;
            }
        }
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i)
            if (i < valid) p[i] = from_f32<T>(v[i]);
    }
}

// vector accesses of I items need this alignment (bytes)
template <typename T, int I> constexpr uintptr_t vec_align() { return (I * sizeof(T)) >= 16 ? 16 : 8; }
template <typename T, int I> __device__ __forceinline__ bool vec_ok(const T *p) {
    return (reinterpret_cast<uintptr_t>(p) & (vec_align<T, I>() - 1)) == 0;
}

// LDS image of a [NB][TC] tile such that lane p's I items are I/4 conflict-free 16-byte reads:
//   element (state n, position p, item i) -> n*TC + (i/4)*(LPR*4) + p*4 + (i%4)
template <int LPR, int I>
__device__ __forceinline__ int tile_off(int n, int p, int i) {
    return n * (LPR * I) + (i >> 2) * (LPR * 4) + p * 4 + (i & 3);
}

// (round 5) the same image with the I/4 quad planes kTilePlanePad<I> floats further apart.  Reads are unchanged (lane p's quad k
// sits at k * plane + 4 p: consecutive lanes, consecutive 16 bytes -- what the four lane groups of ds_read_b128 want).  The
// STAGING writes were the conflict: thread k of stage_bc_tiles owns 4-position group k of a state row = (pos k / (I/4), quad
// k % (I/4)), so consecutive threads wrote whole planes apart.  ds_write_b128 is serviced in contiguous 8-lane groups over 32
// banks (128 bytes: MI355X_MICROARCH.md, LDS); planes LPR * 16 bytes apart are a multiple of that sweep, so the I/4 threads of
// one position shared their banks: a four-way conflict (I = 16) on every write of the tile refill.
// profiles/r04_pmc_sq_scan.txt: SQ_LDS_BANK_CONFLICT = 25 % of SQ_LDS_IDX_ACTIVE for oss_scan_fwd_kernel<bf16,64,16,12>.
// With 128 / I floats of padding per plane a group's 8 threads -- 8 / (I/4) positions x I/4 quads -- cover the 8 slots of the
// sweep once (16 floats were measured first: 25 % -> 10 %, quads 0 / 2 still on one slot).
template <int I> constexpr int kTilePlanePad = 128 / I;
template <int LPR, int I> constexpr int kTileRowPad = LPR * I + (I / 4) * kTilePlanePad<I>;   // floats per state row
template <int LPR, int I>
__device__ __forceinline__ int tile_off_pad(int n, int p, int i) {
    return n * kTileRowPad<LPR, I> + (i >> 2) * (LPR * 4 + kTilePlanePad<I>) + p * 4 + (i & 3);
}

// I items of one lane at scan positions tl .. tl+I-1 of a row of length L.  Forward rows read
// memory tl+i; time-reversed rows read memory L-1-(tl+i) (one contiguous block, mirrored in
// registers).  Positions >= L read as 0.
template <int I, typename T>
__device__ __forceinline__ void load_items_dir(const T *row, int tl, int valid, int L, bool rev, float (&v)[I]) {
    if (!rev) {
        const T *p = row + tl;
        load_items<I>(p, valid, valid == I && vec_ok<T, I>(p), v);
    } else if (valid == I) {
        const T *p = row + (L - tl - I);
        float tmp[I];
        load_items<I>(p, I, vec_ok<T, I>(p), tmp);
#pragma unroll
        for (int i = 0; i < I; ++i) v[i] = tmp[I - 1 - i];
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i) v[i] = (i < valid) ? to_f32(row[L - 1 - tl - i]) : 0.f;
    }
}
template <int I, typename T>
__device__ __forceinline__ void store_items_dir(T *row, int tl, int valid, int L, bool rev, const float (&v)[I]) {
    if (!rev) {
        T *p = row + tl;
        store_items<I>(p, valid, valid == I && vec_ok<T, I>(p), v);
    } else if (valid == I) {
        T *p = row + (L - tl - I);
        float tmp[I];
#pragma unroll
        for (int i = 0; i < I; ++i) tmp[i] = v[I - 1 - i];
        store_items<I>(p, I, vec_ok<T, I>(p), tmp);
    } else {
#pragma unroll
        for (int i = 0; i < I; ++i)
            if (i < valid) row[L - 1 - tl - i] = from_f32<T>(v[i]);
    }
}

// I items of one lane as loaded (not yet converted): 16-bit types I/2 words, float I words
template <typename T, int I> struct RawItems { uint32_t w[I * sizeof(T) / 4]; };

// scan positions tl .. tl+I-1 of a row (memory L-1-t for time-reversed rows); positions >= L read as 0.
// The vector form and the element-wise form are separate functions on purpose: a caller that fetches several rows decides
// ONCE (raw_fast_ok for every pointer) and keeps all the vector loads in one basic block -- a per-load branch makes hipcc
// end every branch with s_waitcnt vmcnt(0), which serialises the loads (one memory round trip each).
template <int I, typename T>
__device__ __forceinline__ const T *raw_block_ptr(const T *row, int tl, int L, bool rev) {
    return rev ? row + (L - tl - I) : row + tl;
}
template <int I, typename T>
__device__ __forceinline__ bool raw_fast_ok(const T *row, int tl, int valid, int L, bool rev) {
    return valid == I && (reinterpret_cast<uintptr_t>(raw_block_ptr<I>(row, tl, L, rev)) & 15u) == 0;
}
template <int I, typename T>
__device__ __forceinline__ RawItems<T, I> load_raw_fast(const T *row, int tl, int L, bool rev) {
    RawItems<T, I> r;
    constexpr int W = I * sizeof(T) / 4;
    const uint32_t *p = reinterpret_cast<const uint32_t *>(raw_block_ptr<I>(row, tl, L, rev));
#pragma unroll
    for (int k = 0; k < W / 4; ++k) {
        const u32x4 q = *reinterpret_cast<const u32x4 *>(p + 4 * k);
        r.w[4 * k] = q.x; r.w[4 * k + 1] = q.y; r.w[4 * k + 2] = q.z; r.w[4 * k + 3] = q.w;
    }
    return r;
}
template <int I, typename T>
__device__ __forceinline__ RawItems<T, I> load_raw_slow(const T *row, int tl, int valid, int L, bool rev) {
    RawItems<T, I> r;
    constexpr int W = I * sizeof(T) / 4;
#pragma unroll
    for (int k = 0; k < W; ++k) r.w[k] = 0u;
#pragma unroll
    for (int i = 0; i < I; ++i) {   // stored in MEMORY order of the block [tl, tl+I) (or its mirror image)
        const int s_ = rev ? (I - 1 - i) : i;   // scan offset held at memory slot i
        if (s_ < valid) {
            const T e = row[rev ? (L - 1 - tl - s_) : (tl + s_)];
            if constexpr (sizeof(T) == 4) r.w[i] = __float_as_uint(e);
            else r.w[i / 2] |= (uint32_t)e.v << (16 * (i & 1));
        }
    }
    return r;
}
template <int I, typename T>
__device__ __forceinline__ RawItems<T, I> load_raw_dir(const T *row, int tl, int valid, int L, bool rev) {
    if (raw_fast_ok<I>(row, tl, valid, L, rev)) return load_raw_fast<I>(row, tl, L, rev);
    return load_raw_slow<I>(row, tl, valid, L, rev);
}
template <int I, typename T>
__device__ __forceinline__ void unpack_raw_dir(const RawItems<T, I> &r, bool rev, float (&v)[I]) {
    float m[I];
    if constexpr (sizeof(T) == 4) {
#pragma unroll
        for (int i = 0; i < I; ++i) m[i] = __uint_as_float(r.w[i]);
    } else {
#pragma unroll
        for (int i = 0; i < I / 2; ++i) unpack2<T>(r.w[i], m[2 * i], m[2 * i + 1]);
    }
#pragma unroll
    for (int i = 0; i < I; ++i) v[i] = rev ? m[I - 1 - i] : m[i];
}

// delta computed where it is consumed (include/vmambair_oss.h: dt_weight): acc[i] = sum_r w[r] * z[r][scan position tl + i],
// r ascending, fp32 -- the dt projection of the archs (MambaSISR6_arch.py:409-411) without its (batch, dim, seqlen) output.
// Split in two so that a kernel can put the loads of ALL rank rows in flight together with its other chunk loads (u, dout):
// every dependent global load is a full memory round trip (~1 us measured for rows another XCD has just written), and a loop
// that loads, converts and accumulates row by row pays it once per row (+70 us on the u:(8,384,4096) backward).
constexpr int kMaxDtRank = 8;
template <typename T, int I> struct DtRows {
    RawItems<T, I> z[kMaxDtRank];
    float w[kMaxDtRank];
};
// all rank rows aligned for the vector form?
template <int I, typename T>
__device__ __forceinline__ bool dt_rows_fast_ok(const T *z0, int64_t rank_stride, int tl, int valid, int L, bool rev) {
    return raw_fast_ok<I>(z0, tl, valid, L, rev) && ((rank_stride * (int64_t)sizeof(T)) & 15) == 0;
}
// every slot is filled (slots past the rank re-read the last row with weight 0): straight-line code whose loads all issue
// before the first wait, and an aggregate the compiler keeps in registers.  `fast` = dt_rows_fast_ok(...), decided by the
// caller together with its other rows.
template <int I, typename T>
__device__ __forceinline__ void dt_rows_load(DtRows<T, I> &d, const T *z0, int64_t rank_stride, const float *w, int R, int tl,
                                             int valid, int L, bool rev, bool fast) {
#pragma unroll
    for (int r = 0; r < kMaxDtRank; ++r) {
        const int rr = min(r, R - 1);
        d.z[r] = fast ? load_raw_fast<I>(z0 + rr * rank_stride, tl, L, rev) : load_raw_slow<I>(z0 + rr * rank_stride, tl, valid, L, rev);
        d.w[r] = (r < R) ? w[rr] : 0.f;
    }
}
template <int I, typename T>
__device__ __forceinline__ void dt_rows_apply(const DtRows<T, I> &d, int R, bool rev, float (&acc)[I]) {
#pragma unroll
    for (int i = 0; i < I; ++i) acc[i] = 0.f;
#pragma unroll
    for (int r = 0; r < kMaxDtRank; ++r) {
        if (r < R) {   // wave-uniform
            float zz[I];
            unpack_raw_dir<I>(d.z[r], rev, zz);
#pragma unroll
            for (int i = 0; i < I; ++i) acc[i] = __builtin_fmaf(d.w[r], zz[i], acc[i]);
        }
    }
}

// Stage nb state rows x TC scan positions of one (batch, group) of B and C into the LDS tiles as
// fp32 (tile_off image).  `rev`: scan position s reads memory L-1-s.  WITH_C = false stages B only (the local pass of
// the time-segmented forward never touches C).
template <typename T, int LPR, int I, int NT, bool WITH_C = true, bool PAD = false>
__device__ __forceinline__ void stage_bc_tiles(float *sB, float *sC, const T *gB, const T *gC, int64_t strideB,
                                               int64_t strideC, int nb, int t0, int L, bool rev, int tid) {
    constexpr int TC = LPR * I;
    constexpr int Q = TC / 4;  // 4-position groups per state row
    const bool fullchunk = (t0 + TC <= L);
    constexpr uintptr_t amask = (sizeof(T) == 4) ? 15u : 7u;
    for (int idx = tid; idx < nb * Q; idx += NT) {
        const int n = idx / Q, k = idx - n * Q;
        const int s0 = t0 + 4 * k;                       // first scan position of the group
        const int m0 = rev ? (L - 4 - s0) : s0;          // lowest memory index of the group
        const T *pb = gB + n * strideB + m0;
        const T *pc = WITH_C ? gC + n * strideC + m0 : pb;
        float b4[4], c4[4] = {0.f, 0.f, 0.f, 0.f};
        if (fullchunk && ((reinterpret_cast<uintptr_t>(pb) | reinterpret_cast<uintptr_t>(pc)) & amask) == 0) {
            if constexpr (sizeof(T) == 4) {
                f32x4 qb = *reinterpret_cast<const f32x4 *>(pb);
#pragma unroll
                for (int j = 0; j < 4; ++j) b4[j] = qb[j];
                if constexpr (WITH_C) {
                    f32x4 qc = *reinterpret_cast<const f32x4 *>(pc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) c4[j] = qc[j];
                }
            } else {
                u32x2 qb = *reinterpret_cast<const u32x2 *>(pb);
                unpack2<T>(qb.x, b4[0], b4[1]); unpack2<T>(qb.y, b4[2], b4[3]);
                if constexpr (WITH_C) {
                    u32x2 qc = *reinterpret_cast<const u32x2 *>(pc);
                    unpack2<T>(qc.x, c4[0], c4[1]); unpack2<T>(qc.y, c4[2], c4[3]);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {  // memory index m0 + j; valid iff its scan position < L
                const int m = m0 + j;
                const bool ok = m >= 0 && m < L;
                b4[j] = ok ? to_f32(gB[n * strideB + m]) : 0.f;
                if constexpr (WITH_C) c4[j] = ok ? to_f32(gC[n * strideC + m]) : 0.f;
            }
        }
        f32x4 vb, vc;
        if (rev) { vb = f32x4{b4[3], b4[2], b4[1], b4[0]}; vc = f32x4{c4[3], c4[2], c4[1], c4[0]}; }
        else     { vb = f32x4{b4[0], b4[1], b4[2], b4[3]}; vc = f32x4{c4[0], c4[1], c4[2], c4[3]}; }
        const int pos = (4 * k) / I, i0 = (4 * k) % I;
        const int off = PAD ? tile_off_pad<LPR, I>(n, pos, i0) : tile_off<LPR, I>(n, pos, i0);
        *reinterpret_cast<f32x4 *>(sB + off) = vb;
        if constexpr (WITH_C) *reinterpret_cast<f32x4 *>(sC + off) = vc;
    }
}

template <typename T> __device__ __forceinline__ bool aligned16(const T *p) {
    return (reinterpret_cast<uintptr_t>(p) & 15u) == 0;
}

// ---------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float exp2_hw(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32

// softplus with the reference's threshold (selective_scan_fwd_kernel.cuh:115-118):
//   x <= 20 ? log1p(exp(x)) : x.   `e` returns exp(x) for the backward's sigmoid.
// log1p(e) is evaluated as log(1+e) * e / ((1+e) - 1), which keeps full relative accuracy for
// small e with two transcendentals and one reciprocal.
__device__ __forceinline__ float softplus_thr(float x, float &e) {
    e = exp2_hw(x * kLog2e);
    const float s = 1.0f + e;
    const float d = s - 1.0f;
    const float l = __builtin_amdgcn_logf(s) * kLn2;  // v_log_f32 is log2
    const float r = (d == 0.f) ? e : l * (e * __builtin_amdgcn_rcpf(d));
    return (x <= 20.f) ? r : x;
}

// ---------------------------------------------------------------------------------------------
// DPP helpers (gfx9 encodings)
// ---------------------------------------------------------------------------------------------
constexpr int kDppRowShr1 = 0x111, kDppRowShr2 = 0x112, kDppRowShr4 = 0x114, kDppRowShr8 = 0x118;
constexpr int kDppWaveShl1 = 0x130, kDppWaveShr1 = 0x138;
constexpr int kDppRowBcast15 = 0x142, kDppRowBcast31 = 0x143;

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL,
                                                      ROW_MASK, 0xf, false));
}

// One Kogge-Stone step of the inclusive scan of the linear-recurrence monoid
//   (P_l, h_l) o (P_r, h_r) = (P_r P_l, P_r h_l + h_r)        [left = earlier]
// (reference operator: SSMScanOp, selective_scan_common.h:89-96).  Lanes without a source keep
// the identity (1, 0), so the step is a no-op for them.
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ void scan_step(float &P, float &h) {
    const float hl = dpp_f32<CTRL, ROW_MASK>(0.f, h);
    const float Pl = dpp_f32<CTRL, ROW_MASK>(1.f, P);
    h = __builtin_fmaf(P, hl, h);
    P = P * Pl;
}

// Hand-scheduled form of scan_step: on gfx9-family ISAs VOP2 instructions take a DPP source, and a
// lane whose DPP source is out of range (or whose row is masked) is simply not written -- exactly
// the identity behaviour the scan needs -- so one step is two VALU instructions
//     h += dpp(h) * P ;  P *= dpp(P)
// instead of the six the builtin form compiles to.  Wait states: a VALU write followed by a DPP
// read of the same VGPR needs two intervening states (s_nop 1 up front, s_nop 0 between steps).
#define OSS_DPP_STEP(CTRL)                                   \
    "v_fmac_f32_dpp %0, %0, %1 " CTRL " bank_mask:0xf\n\t" \
    "v_mul_f32_dpp %1, %1, %1 " CTRL " bank_mask:0xf\n\t"
template <int LPR>
__device__ __forceinline__ void segment_scan(float &P, float &h) {
    if constexpr (LPR == 16) {
        asm volatile("s_nop 1\n\t" OSS_DPP_STEP("row_shr:1 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:2 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:4 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:8 row_mask:0xf")
                     : "+v"(h), "+v"(P));
    } else if constexpr (LPR == 32) {
        asm volatile("s_nop 1\n\t" OSS_DPP_STEP("row_shr:1 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:2 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:4 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:8 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_bcast:15 row_mask:0xa")
                     : "+v"(h), "+v"(P));
    } else {
        asm volatile("s_nop 1\n\t" OSS_DPP_STEP("row_shr:1 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:2 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:4 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_shr:8 row_mask:0xf") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_bcast:15 row_mask:0xa") "s_nop 0\n\t"
                     OSS_DPP_STEP("row_bcast:31 row_mask:0xc")
                     : "+v"(h), "+v"(P));
    }
}
#undef OSS_DPP_STEP
// Four independent (P, h) pairs scanned together over the 64 lanes: the same steps as segment_scan<64>, interleaved, so that the
// pairs fill each other's DPP wait states and latencies (oss_channel.hip: the four states of a lane group).
#define OSS_DPP_STEP4(CTRL)                                    \
    "v_fmac_f32_dpp %0, %0, %4 " CTRL " bank_mask:0xf\n\t"   \
    "v_fmac_f32_dpp %1, %1, %5 " CTRL " bank_mask:0xf\n\t"   \
    "v_fmac_f32_dpp %2, %2, %6 " CTRL " bank_mask:0xf\n\t"   \
    "v_fmac_f32_dpp %3, %3, %7 " CTRL " bank_mask:0xf\n\t"   \
    "v_mul_f32_dpp %4, %4, %4 " CTRL " bank_mask:0xf\n\t"    \
    "v_mul_f32_dpp %5, %5, %5 " CTRL " bank_mask:0xf\n\t"    \
    "v_mul_f32_dpp %6, %6, %6 " CTRL " bank_mask:0xf\n\t"    \
    "v_mul_f32_dpp %7, %7, %7 " CTRL " bank_mask:0xf\n\t"
__device__ __forceinline__ void segment_scan4_64(float (&P)[4], float (&h)[4]) {
    asm volatile("s_nop 1\n\t" OSS_DPP_STEP4("row_shr:1 row_mask:0xf") OSS_DPP_STEP4("row_shr:2 row_mask:0xf")
                 OSS_DPP_STEP4("row_shr:4 row_mask:0xf") OSS_DPP_STEP4("row_shr:8 row_mask:0xf")
                 OSS_DPP_STEP4("row_bcast:15 row_mask:0xa") OSS_DPP_STEP4("row_bcast:31 row_mask:0xc")
                 : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(P[0]), "+v"(P[1]), "+v"(P[2]), "+v"(P[3]));
}
#undef OSS_DPP_STEP4

// Builtin (compiler-scheduled) form of the same scan; kept as the readable definition and used by
// the unit check that both forms agree.
template <int LPR>
__device__ __forceinline__ void segment_scan_builtin(float &P, float &h) {
    scan_step<kDppRowShr1>(P, h);
    scan_step<kDppRowShr2>(P, h);
    scan_step<kDppRowShr4>(P, h);
    scan_step<kDppRowShr8>(P, h);
    if constexpr (LPR >= 32) scan_step<kDppRowBcast15, 0xa>(P, h);
    if constexpr (LPR >= 64) scan_step<kDppRowBcast31, 0xc>(P, h);
}

// sum over segments of LPR lanes; every lane of the segment's LAST position holds the total.
template <int LPR>
__device__ __forceinline__ float segment_sum_to_last(float v) {
    v += dpp_f32<kDppRowShr1>(0.f, v);
    v += dpp_f32<kDppRowShr2>(0.f, v);
    v += dpp_f32<kDppRowShr4>(0.f, v);
    v += dpp_f32<kDppRowShr8>(0.f, v);
    if constexpr (LPR >= 32) v += dpp_f32<kDppRowBcast15, 0xa>(0.f, v);
    if constexpr (LPR >= 64) v += dpp_f32<kDppRowBcast31, 0xc>(0.f, v);
    return v;
}

// value of lane-1 (whole wave); lane 0 and, through `seg_first`, every segment's first lane get `fill`.
__device__ __forceinline__ float shift_from_prev_lane(float v, float fill, bool seg_first) {
    const float s = dpp_f32<kDppWaveShr1>(fill, v);
    return seg_first ? fill : s;
}
// value of lane+1; the segment's last lane gets `fill`.
__device__ __forceinline__ float shift_from_next_lane(float v, float fill, bool seg_last) {
    const float s = dpp_f32<kDppWaveShl1>(fill, v);
    return seg_last ? fill : s;
}

// mirror a value inside segments of LPR lanes (lane p <-> LPR-1-p)
template <int LPR>
__device__ __forceinline__ float segment_mirror(float v, int lane) {
    const int src = (lane & ~(LPR - 1)) | ((LPR - 1) - (lane & (LPR - 1)));
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src << 2, __float_as_int(v)));
}


}  // namespace oss
