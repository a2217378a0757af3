// torch's CPU `sum` of a contiguous vector in ITS order, for vectors of any length, on the whole chip.
//
// Reference: the loss of MSEFast is `(pred - tgt).abs().pow(2).mean()` (quantization/observer.py:420-432) and the LSQ / LSQ+
// parameter gradients are `sum_to_size` reductions of autograd (quantization/util_quant.py:29-67); both are torch.sum of
// a contiguous CPU tensor = ATen's cascade_sum / vectorized_inner_sum (aten/src/ATen/native/cpu/SumKernel.cpp; restated
// and pinned against torch.sum in oracle/aten_sum.py).  With torch on ONE thread that order is defined for every length:
//
//   * the vector is read as n / W SIMD vectors of W lanes; vector i belongs to interleaved accumulator i % 4 -- so the
//     first size * 4W elements (size = n / W / 4) form `size` ROWS of NC = 4W independent COLUMNS, element (row, col) at
//     row * NC + col;
//   * every column is added by a four-level cascade with step S = 2^P, P = max(4, ceil_log2(size) / 4): level 0 adds S
//     consecutive rows, is added to level 1 and cleared; level 1 is added to level 2 after S level-0 flushes (S^2 rows),
//     level 2 to level 3 after S^3 rows; when the rows run out the levels are added to level 0 in order 1, 2, 3;
//   * the n / W % 4 left-over vectors are added to accumulator 0, accumulators 1, 2, 3 to accumulator 0 in that order,
//     then the n % W trailing scalars one by one to a scalar that starts at 0, then the W lanes in order.
//
// A level-0 BLOCK (S rows), a level-1 CHUNK (S blocks) and a level-2 UNIT (S chunks) are left-to-right sums that start
// from zero and depend on nothing else: that is the parallelism.  Stage 1 (`cascade_units`, every workgroup): a
// workgroup takes chunks; thread (block, column) adds its S rows in order, the block sums meet in LDS, thread (column)
// adds the S block sums in order and publishes the chunk's NC column sums.  Stage 2 (`cascade_finish`, the workgroup that
// arrives last): thread (unit, column) adds S chunk sums in order, thread (column) adds the unit sums in order (level 3),
// then the open levels, and one thread folds the columns, left-overs, tail and lanes.  Same additions, same order, as the
// one-thread CPU kernel -- for any n; the elements are produced on the fly by `term(e, out[NS])` (NS sums share a pass:
// the LSQ+ backward has four), called exactly once per element over both stages.
//
// Scratch (caller's): NS * (chunks + 2) * NC values of T, see cascade_scratch_bytes.  P <= 5 (n < 2^25 * NC).
#pragma once
#include "osq_device.h"

namespace osq {

struct CascadeGeom {
    int64_t n, n_vec, size, blocks, chunks;
    int W, NC, P, S, tail_blocks, tail_rows;
};

__host__ __device__ inline int cascade_ceil_log2(int64_t x) {
    int p = 0;
    while ((static_cast<int64_t>(1) << p) < x) ++p;
    return p;
}

__host__ __device__ inline CascadeGeom cascade_geom(int64_t n, int W) {
    CascadeGeom g;
    g.n = n;
    g.W = W;
    g.NC = 4 * W;
    g.n_vec = n >> cascade_ceil_log2(W);                   // W is 4, 8 or 16: a shift, not a 64-bit division per workgroup
    g.size = g.n_vec / 4;
    const int p = cascade_ceil_log2(g.size) / 4;
    g.P = p < 4 ? 4 : p;
    g.S = 1 << g.P;
    g.blocks = g.size >> g.P;
    g.chunks = g.blocks >> g.P;
    g.tail_blocks = static_cast<int>(g.blocks - (g.chunks << g.P));
    g.tail_rows = static_cast<int>(g.size - (g.blocks << g.P));
    return g;
}

// bytes of scratch one ordered sum of n elements (NS sums, values of elem_size bytes, W lanes) publishes
inline size_t cascade_scratch_bytes(int64_t n, int W, int ns, size_t elem_size) {
    const CascadeGeom g = cascade_geom(n, W);
    return static_cast<size_t>(ns) * static_cast<size_t>(g.chunks + 2) * static_cast<size_t>(g.NC) * elem_size;
}
constexpr int kCascadeMaxP = 5;

template <typename T> __device__ __forceinline__ void cascade_publish(T* p, T v);
template <> __device__ __forceinline__ void cascade_publish<float>(float* p, float v) { publish_f32(p, v); }
template <> __device__ __forceinline__ void cascade_publish<double>(double* p, double v) { publish_f64(p, v); }

// sc1 loads through a buffer descriptor: agent-scope like consume_f32 / consume_f64, but plain loads to the compiler, so
// the S loads of a chain are all in flight before the first addition (ordered atomic loads are consumed one round trip at
// a time)
typedef unsigned int cascade_v2u32 __attribute__((ext_vector_type(2)));
struct CascadeReader {
    __amdgpu_buffer_rsrc_t rsrc;
    __device__ __forceinline__ CascadeReader(const void* base, size_t bytes)
        : rsrc(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000)) {}
    __device__ __forceinline__ float get_f32(int64_t idx) const {
        return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, static_cast<unsigned int>(idx * 4), 0, 16));
    }
    __device__ __forceinline__ double get_f64(int64_t idx) const {
        const cascade_v2u32 w = __builtin_amdgcn_raw_buffer_load_b64(rsrc, static_cast<unsigned int>(idx * 8), 0, 16);
        return __longlong_as_double(static_cast<long long>((static_cast<unsigned long long>(w.y) << 32) | w.x));
    }
    template <typename T> __device__ __forceinline__ T get(int64_t idx) const;
};
template <> __device__ __forceinline__ float CascadeReader::get<float>(int64_t idx) const { return get_f32(idx); }
template <> __device__ __forceinline__ double CascadeReader::get<double>(int64_t idx) const { return get_f64(idx); }

// ---- stage 1: every workgroup; `lds` holds NS * S * NC values of T
// bid / nblk: this workgroup's index among the workgroups that add THIS vector (the whole grid unless a launch serves several)
template <typename T, int NS, int THREADS, typename Term>
__device__ __forceinline__ void cascade_units(const CascadeGeom& g, T* __restrict__ part, T* lds, Term term, const unsigned int bid,
                                              const unsigned int nblk_grid, const int64_t first_unit = 0) {
    const int64_t units = g.chunks + 1;                     // the last one is what the levels still hold at the end
    const int64_t sstride = (g.chunks + 2) * g.NC;          // part[s][m][c]; m == chunks: open level 1, chunks + 1: open level 0
    const int nc_shift = __builtin_ctz(static_cast<unsigned int>(g.NC));
    for (int64_t m = first_unit + bid; m < units; m += nblk_grid) {
        const bool full = m < g.chunks;
        const int nblk = full ? g.S : g.tail_blocks;
        const int ntask = (nblk + ((!full && g.tail_rows) ? 1 : 0)) << nc_shift;
        for (int task = threadIdx.x; task < ntask; task += THREADS) {
            const int blk = task >> nc_shift, c = task & (g.NC - 1);
            const bool open0 = blk >= nblk;                 // the rows behind the last full block: they stay in level 0
            const int64_t row0 = open0 ? (g.blocks << g.P) : (((m << g.P) + blk) << g.P);
            const int nrows = open0 ? g.tail_rows : g.S;
            T acc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = T(0);
            // the S terms of a block are independent of each other (only their ADDITION is ordered): KB of them are
            // produced side by side -- their loads in flight together, their division chains interleaved -- then added in
            // order.  (Produced and added one by one, every row waited for its own load: 15 us per evaluation of a
            // [32,128,768] site instead of the few its 7 MB cost.)
            constexpr int KB = NS == 1 ? 16 : 8;
            int r = 0;
            for (; r + KB <= nrows; r += KB) {
                T t[KB][NS];
#pragma unroll
                for (int j = 0; j < KB; ++j) term(((row0 + r + j) << nc_shift) + c, t[j]);
#pragma unroll
                for (int j = 0; j < KB; ++j)
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[s] = acc[s] + t[j][s];
            }
            for (; r < nrows; ++r) {
                T t[NS];
                term(((row0 + r) << nc_shift) + c, t);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s] = acc[s] + t[s];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (open0) cascade_publish<T>(&part[s * sstride + ((g.chunks + 1) << nc_shift) + c], acc[s]);
                else lds[(((s << g.P) + blk) << nc_shift) + c] = acc[s];
            }
        }
        __syncthreads();
        for (int t = threadIdx.x; t < (NS << nc_shift); t += THREADS) {
            const int s = t >> nc_shift, c = t & (g.NC - 1);
            T acc = T(0);
            for (int b = 0; b < nblk; ++b) acc = acc + lds[(((s << g.P) + b) << nc_shift) + c];
            cascade_publish<T>(&part[s * sstride + (m << nc_shift) + c], acc);
        }
        // Every wave drains ITS published values before the barrier: the workgroup's ticket (grid_last_block, thread 0)
        // only waits for thread 0's own wave, and a workgroup-scope barrier is no promise about stores still in flight
        // to memory (MI355X_MICROARCH.md, "valid forms": sc1 payload -> vmcnt(0) -> flag).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
}

template <typename T, int NS, int THREADS, typename Term>
__device__ __forceinline__ void cascade_units(const CascadeGeom& g, T* __restrict__ part, T* lds, Term term) {
    cascade_units<T, NS, THREADS, Term>(g, part, lds, term, blockIdx.x, gridDim.x);
}

// A barrier for exchanges through LDS only.  __syncthreads() is a workgroup-scope fence around s_barrier: it also waits for every
// global load and store the wave has in flight (vmcnt(0)).
__device__ __forceinline__ void cascade_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// cur <- nxt of the pipelined stage.  For a float the copy is an explicit v_mov: with a plain assignment the compiler may
// alias `cur` to the load's destination registers, and its wait-count pass then re-waits (vmcnt(0): the NEXT step's loads too)
// at every basic block that reads them -- seen in the S = 32 form, whose second step began behind the loads it had just issued.
template <typename Raw> __device__ __forceinline__ void cascade_rotate(Raw& cur, const Raw& nxt) { cur = nxt; }
template <> __device__ __forceinline__ void cascade_rotate<float>(float& cur, const float& nxt) {
    asm volatile("v_mov_b32 %0, %1" : "=v"(cur) : "v"(nxt));
}

// Stage 1 of ONE sum over the FULL level-1 chunks with the memory latency hidden.  The generic form above is a chain per
// chunk -- loads, terms, barrier, the S block sums, publish -- of ~3 us with two workgroups per CU to overlap it: 1.7 us per
// million elements, the whole cost of a strict MSEFast round.  Here a thread's S raw inputs of its NEXT chunk (`load(e)`,
// Raw = what `eval` needs of element e) are in flight while the current chunk's terms are computed (`eval(raw)` -> T) and
// added in order; the block sums go through a double-buffered LDS tile (one barrier per chunk), and a workgroup whose
// threads outnumber a chunk's S * NC (block, column) pairs takes G = THREADS / (S * NC) chunks at a time (float64: NC is
// half as wide).  NS sums share the pass (`eval(raw, e, t[NS])`; it may write per-element results on the way).  Same
// additions in the same order.  Requires S * NC <= THREADS and chunks > 0.
// The open unit (m == chunks) stays with cascade_units(..., first_unit = chunks).
//
// What keeps the loads in flight (round 5, read off the ISA and stamped):
//  * the fetch is BRANCH-FREE -- behind a condition the compiler has no count of the loads in flight at the join and puts
//    `s_waitcnt vmcnt(0)` in front of the first term: the current group's arithmetic then waits for the NEXT group's loads;
//  * the barrier of a group is an LDS-only barrier, not __syncthreads() (which drains vmcnt: the next group's loads);
//  * NO STORE leaves inside the loop over a batch's groups: vmcnt counts loads and stores in one order, so the wave that
//    published a group's block sums (write-through, ~1.5 us to acknowledge) stalled on them at its next wait for loads,
//    and the workgroup on that wave at the next barrier -- and a store behind a condition costs the compiler its count again.
//    The block sums of a BATCH of up to kCascadeHold groups wait in LDS (`lds` beyond the two tiles; lds_values is its
//    capacity, >= 2 * NS * THREADS + NS * G * NC) and leave in one burst of stores from as many lanes.
// NCS >= 0: the caller knows log2(NC) at compile time (it branched on it): a thread's R loads are then ONE 64-bit address
// and R immediate offsets (k * NC elements apart), not R address computations.
constexpr int kCascadeHold = 16;
// LDS values (of T) cascade_chunks_pipelined<T, NS, P, THREADS> needs at least: two tiles + one group's block sums
template <int NS, int P, int THREADS>
constexpr bool cascade_lds_fits(const int lds_values) { return lds_values >= 2 * NS * THREADS + NS * (THREADS >> P); }
// EVALW = 4 (NS = 1): eval(raw[4], t[4]) produces four terms of a thread's rows side by side -- a term with a rare slow form then
// takes ONE branch per four elements instead of four (msefast.hip: the lean float64 term's tie guard).
template <typename T, int NS, int P, int THREADS, typename Raw, typename Load, typename Eval, int NCS = -1, int EVALW = 1>
__device__ __forceinline__ void cascade_chunks_pipelined(const CascadeGeom& g, T* __restrict__ part, T* lds, Load load, Eval eval,
                                                         const unsigned int bid, const unsigned int nblk_grid, const int lds_values) {
    constexpr int S = 1 << P;
    constexpr int R = 16;                                   // rows in flight per thread: a block of S = 32 rows is two steps
    constexpr int H = S / R;
    constexpr int KB = NS == 1 ? 16 : 8;                    // terms computed side by side
    const int nc_shift = NCS >= 0 ? NCS : __builtin_ctz(static_cast<unsigned int>(g.NC));
    const int64_t sstride = (g.chunks + 2) * g.NC;          // part[s][m][c], as in cascade_units
    const int tpc = S << nc_shift;                          // (block, column) pairs of a chunk
    const int G = THREADS / tpc;
    const int64_t ngroups = (g.chunks + G - 1) / G;
    const int tid = threadIdx.x;
    const int j = tid / tpc, rem = tid - j * tpc, blk = rem >> nc_shift, c = rem & (g.NC - 1);
    const bool lane_ok = tid < G * tpc;
    const int cols = G << nc_shift;                         // block sums a group leaves, per sum
    // block sums held back in LDS behind the two tiles: pub[hold][NS * cols]
    const int room = (lds_values - 2 * NS * THREADS) / (NS * cols);
    // pub[] lives BEHIND the two tiles: an array that ends before one group's block sums fit (the old contract asked for the
    // tiles only) would be written past its end.  cols <= THREADS / S, so cascade_lds_fits<...>() -- which every caller
    // static_asserts on its compile-time capacity -- implies room >= 1; the trap is for a caller that did not.
    if (room < 1) __builtin_trap();
    // NS > 1 is the LSQ+ backward, whose eval stores dx element by element: its loop is not store-free whatever happens to the
    // block sums, and a batch of one measured 2-8 % ahead there (same-box A/B)
    const int hold = NS > 1 ? 1 : (room > kCascadeHold ? kCascadeHold : (room < 1 ? 1 : room));
    T* const pub = lds + 2 * NS * THREADS;
    Raw cur[R], nxt[R];
    // a lane with nothing to fetch (past the last chunk, no next group) reads rows 0..R-1 of chunk 0 instead: every such lane
    // of a column the same 16 lines, in bounds because chunks > 0, never used
    auto fetch = [&](const int64_t grp, const int h, Raw (&r)[R]) {
        const int64_t m = grp * G + j;
        const int64_t e0 = (lane_ok && m < g.chunks) ? (((((m << P) + blk) << P) + h * R) << nc_shift) + c : static_cast<int64_t>(c);
#pragma unroll
        for (int k = 0; k < R; ++k) r[k] = load(e0 + (static_cast<int64_t>(k) << nc_shift));
    };
    int64_t grp = bid;
    if (grp < ngroups) fetch(grp, 0, cur);
    int buf = 0;
    while (grp < ngroups) {
        // ---- a batch of up to `hold` groups: loads, arithmetic and LDS only
        const int64_t grp0 = grp;
        int held = 0;
        for (; grp < ngroups && held < hold; grp += nblk_grid, ++held) {
            const int64_t m = grp * G + j;
            T* const tile = lds + buf * (NS * THREADS);
            T acc[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s] = T(0);
#pragma unroll
            for (int h = 0; h < H; ++h) {
                if (h + 1 < H) fetch(grp, h + 1, nxt);
                else fetch(grp + nblk_grid < ngroups ? grp + nblk_grid : ngroups, 0, nxt);     // ngroups: no such chunk, the dummy rows
                if (lane_ok && m < g.chunks) {
                    const int64_t row0 = (((m << P) + blk) << P) + h * R;
#pragma unroll
                    for (int k0 = 0; k0 < R; k0 += KB) {
                        T t[KB][NS];
                        if constexpr (EVALW == 4) {
                            static_assert(NS == 1 && KB % 4 == 0, "EVALW = 4: one sum, terms in fours");
#pragma unroll
                            for (int k = 0; k < KB; k += 4) {
                                const Raw r4[4] = {cur[k0 + k], cur[k0 + k + 1], cur[k0 + k + 2], cur[k0 + k + 3]};
                                T t4[4];
                                eval(r4, t4);
                                t[k][0] = t4[0]; t[k + 1][0] = t4[1]; t[k + 2][0] = t4[2]; t[k + 3][0] = t4[3];
                            }
                        } else {
#pragma unroll
                            for (int k = 0; k < KB; ++k) eval(cur[k0 + k], ((row0 + k0 + k) << nc_shift) + c, t[k]);
                        }
#pragma unroll
                        for (int k = 0; k < KB; ++k)
#pragma unroll
                            for (int s = 0; s < NS; ++s) acc[s] = acc[s] + t[k][s];
                    }
                }
#pragma unroll
                for (int k = 0; k < R; ++k) {
                    if constexpr (H > 1) cascade_rotate<Raw>(cur[k], nxt[k]);       // measured: S = 32 a round 150 -> 138 us, S = 16 the plain copy 4 % ahead
                    else cur[k] = nxt[k];
                }
            }
            if (lane_ok && m < g.chunks) {
#pragma unroll
                for (int s = 0; s < NS; ++s) tile[s * THREADS + (((j << P) + blk) << nc_shift) + c] = acc[s];
            }
            cascade_lds_barrier();                          // the tile is complete (and the previous batch's burst has read pub)
            for (int t = tid; t < NS * cols; t += THREADS) {
                const int s = t / cols, r = t - s * cols;
                const int jj = r >> nc_shift, cc = r & (g.NC - 1);
                T v[S];
#pragma unroll
                for (int b = 0; b < S; ++b) v[b] = tile[s * THREADS + (((jj << P) + b) << nc_shift) + cc];
                T a = T(0);
#pragma unroll
                for (int b = 0; b < S; ++b) a = a + v[b];
                pub[held * (NS * cols) + t] = a;            // a chunk past the last one: whatever the tile held, never published
            }
            buf ^= 1;                                       // the next chunk's block sums go to the other tile: one barrier per chunk
        }
        // ---- the batch's burst of stores: its block sums, from as many lanes
        cascade_lds_barrier();
        for (int t = tid; t < held * NS * cols; t += THREADS) {
            const int hslot = t / (NS * cols), q = t - hslot * (NS * cols);
            const int s = q / cols, r = q - s * cols;
            const int jj = r >> nc_shift, cc = r & (g.NC - 1);
            const int64_t mm = (grp0 + static_cast<int64_t>(hslot) * nblk_grid) * G + jj;
            if (mm < g.chunks) cascade_publish<T>(&part[s * sstride + (mm << nc_shift) + cc], pub[t]);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // every wave's published sums have left before the workgroup's ticket
    __syncthreads();
}

// Lanes of ONE wave exchange through LDS: a wave's LDS operations reach the LDS in program order, so no s_barrier is needed --
// the fences keep the compiler from moving them across the hand-over.
__device__ __forceinline__ void cascade_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- stage 2: the last workgroup; `lds` holds lds_values values of T (>= NS * NC + NC); the sums arrive in out[] of thread 0
template <typename T, int NS, int THREADS, typename Term>
__device__ __forceinline__ void cascade_finish(const CascadeGeom& g, const T* __restrict__ part, T* lds, int lds_values, Term term,
                                               T (&out)[NS]) {
    if (g.n < g.W) {                                        // scalar_inner_sum: four interleaved scalars, no cascade below 64 elements
        if (threadIdx.x == 0) {
            T acc[NS][4];
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int k = 0; k < 4; ++k) acc[s][k] = T(0);
            const int size = static_cast<int>(g.n / 4);
            for (int i = 0; i < size; ++i)
                for (int k = 0; k < 4; ++k) {
                    T t[NS];
                    term(i * 4 + k, t);
#pragma unroll
                    for (int s = 0; s < NS; ++s) acc[s][k] = acc[s][k] + t[s];
                }
            for (int64_t i = static_cast<int64_t>(size) * 4; i < g.n; ++i) {
                T t[NS];
                term(i, t);
#pragma unroll
                for (int s = 0; s < NS; ++s) acc[s][0] = acc[s][0] + t[s];
            }
#pragma unroll
            for (int s = 0; s < NS; ++s) out[s] = ((acc[s][0] + acc[s][1]) + acc[s][2]) + acc[s][3];
        }
        return;
    }
    const int nc_shift = __builtin_ctz(static_cast<unsigned int>(g.NC));
    const int64_t sstride = (g.chunks + 2) * g.NC;
    const int64_t units2 = g.chunks >> g.P;                 // full level-2 units
    const CascadeReader rd(part, static_cast<size_t>(NS) * static_cast<size_t>(sstride) * sizeof(T));
    T* const col = lds;                                     // [NS][NC]
    T* const tile = lds + (NS << nc_shift);                 // [NS][QT][NC]: the NS sums advance together
    const int QT = ((lds_values - (NS << nc_shift)) / NS) >> nc_shift;
    const int tid = threadIdx.x;
    const bool owner = tid < (NS << nc_shift);              // thread (s, c): level 3 and the open levels of column c of sum s
    const int my_s = tid >> nc_shift, my_c = tid & (g.NC - 1);
    T acc3 = T(0);
    // the open levels' loads (up to 16 chunk sums behind the last full unit, the open level-1 and level-0 sums) leave with
    // the first level-2 loads instead of after them: one round trip to memory less on the serial tail of every evaluation
    const int64_t open_base = my_s * sstride, open_m0 = units2 << g.P;
    T open_v[16], open_a1 = T(0), open_a0 = T(0);
    if (owner) {
#pragma unroll
        for (int j = 0; j < 16; ++j) open_v[j] = (open_m0 + j < g.chunks) ? rd.get<T>(open_base + ((open_m0 + j) << nc_shift) + my_c) : T(0);
        open_a1 = rd.get<T>(open_base + (g.chunks << nc_shift) + my_c);
        open_a0 = g.tail_rows ? rd.get<T>(open_base + ((g.chunks + 1) << nc_shift) + my_c) : T(0);
    }
    for (int64_t q0 = 0; q0 < units2; q0 += QT) {
        const int nq = static_cast<int>(units2 - q0 < QT ? units2 - q0 : QT);
        const int per_sum = nq << nc_shift;
        for (int task = tid; task < NS * per_sum; task += THREADS) {
            const int s = task / per_sum, r = task - s * per_sum;
            const int q = r >> nc_shift, c = r & (g.NC - 1);
            const int64_t first = s * sstride + ((((q0 + q) << g.P)) << nc_shift) + c;
            T a = T(0);
            for (int j0 = 0; j0 < g.S; j0 += 16) {
                T v[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = rd.get<T>(first + (static_cast<int64_t>(j0 + j) << nc_shift));
#pragma unroll
                for (int j = 0; j < 16; ++j) a = a + v[j];
            }
            tile[((s * QT + q) << nc_shift) + c] = a;
        }
        __syncthreads();
        if (owner)
            for (int q = 0; q < nq; ++q) acc3 = acc3 + tile[((my_s * QT + q) << nc_shift) + my_c];
        __syncthreads();
    }
    if (owner) {
        const int64_t base = open_base;
        T acc2 = T(0);
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (open_m0 + j < g.chunks) acc2 = acc2 + open_v[j];
        for (int64_t m0 = open_m0 + 16; m0 < g.chunks; m0 += 16) {      // S = 32: up to 15 more chunk sums, loaded side by side, added in order
            T v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = (m0 + j < g.chunks) ? rd.get<T>(base + ((m0 + j) << nc_shift) + my_c) : T(0);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (m0 + j < g.chunks) acc2 = acc2 + v[j];
        }
        const T acc1 = open_a1;
        T acc0 = open_a0;
        acc0 = acc0 + acc1;
        acc0 = acc0 + acc2;
        acc0 = acc0 + acc3;
        col[(my_s << nc_shift) + my_c] = acc0;
    }
    __syncthreads();
    // the elements behind the rows: n / W % 4 left-over vectors and n % W trailing scalars, fewer than NC together.  Their
    // terms are produced by as many threads side by side (one load latency instead of up to NC - 1 in a row) into the
    // tile, which stage 2 no longer needs; thread 0 adds them in the kernel's order.
    const int64_t rest0 = g.size * 4 * g.W;
    const int nrest = static_cast<int>(g.n - rest0);
    if (tid < nrest) {
        T t[NS];
        term(rest0 + tid, t);
#pragma unroll
        for (int s = 0; s < NS; ++s) tile[(s << nc_shift) + tid] = t[s];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int nvec_rest = static_cast<int>(g.n_vec - g.size * 4);
        for (int v = 0; v < nvec_rest; ++v)                 // left-over vectors join accumulator 0
            for (int l = 0; l < g.W; ++l)
#pragma unroll
                for (int s = 0; s < NS; ++s) col[(s << nc_shift) + l] = col[(s << nc_shift) + l] + tile[(s << nc_shift) + v * g.W + l];
        T fin[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) fin[s] = T(0);
        for (int k = nvec_rest * g.W; k < nrest; ++k)       // trailing scalars
#pragma unroll
            for (int s = 0; s < NS; ++s) fin[s] = fin[s] + tile[(s << nc_shift) + k];
        for (int l = 0; l < g.W; ++l)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const T* cs = col + (s << nc_shift);
                const T lane = ((cs[l] + cs[g.W + l]) + cs[2 * g.W + l]) + cs[3 * g.W + l];
                fin[s] = fin[s] + lane;
            }
#pragma unroll
        for (int s = 0; s < NS; ++s) out[s] = fin[s];
    }
}

}  // namespace osq
