// gfx950 (MI355X / CDNA4) kernels of the content-defined chunker + SHA-256 engine.
//
// Hot path = the chunk loop behind transfer.ArchiveWriter.WriteEntryReader
// (reference internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836),
// i.e. the Buzhash scan + per-chunk SHA-256 of github.com/pbs-plus/pxar v0.34.0
// parameterised by buzhash.NewConfig (commit_orchestrate.go:143-149). Decomposition
// (DESIGN.md): (1) candidate scan — every stream position's 64-byte window hash is a
// pure function of those 64 bytes once chunk_size >= 64, so all positions are tested in
// parallel, HBM-bound; (2) compaction + min/max resolution over the sparse candidate
// list; (3) one SHA-256 lane per chunk, lanes pulling chunks from a queue, VALU-bound.
// Integer/byte work only: no MFMA anywhere.
#include "kernels.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include <rocprim/device/device_radix_sort.hpp>

namespace pbsk {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================
// (1) Buzhash candidate scan
// =====================================================================================
// h(i) = XOR_{k=0..63} rotl(T[b[i-k]], k mod 32): a pure function of the last 64 bytes, so every position is tested
// independently. The table is pre-rotated by r = 32-bits on the host: rotation commutes with the recurrence, and
// (h & mask) >= break_min becomes the single unsigned compare rotl(h,r) >= break_min << r, so no AND is needed per byte.
// (Rounds 1-5 kept two earlier kernels selectable for A/B runs — an LDS-tiled one, 2.28 TB/s, and a per-lane register
// streaming one, 4.36 TB/s: docs/NOTES.md. Gone with round 6: k_scan3 below is the scan.)
// -------------------------------------------------------------------------------------
// Candidate scan, cooperative-load form. Per-lane streaming of whole lines is bound by the texture addresser
// (64 distinct 128-byte lines per load instruction: 4.14 TB/s for pure loads, profiles/
// r01_ubench_load_pattern_ceiling.log), so here the four lanes of a quad fetch 64 CONTIGUOUS bytes of one
// strip per instruction (16 lines per instruction; 6.0 TB/s ceiling) and a small per-wave LDS stage
// transposes each 64-byte half-line back to "one lane = one strip": 4 ds_write_b128 at
// [strip][piece] with an 80-byte strip pitch, 4 ds_read_b128 of the lane's own row (pitch 5 slots, odd:
// conflict-free). A half-line is exactly one revolution of the 64-entry window ring, so ring indices
// stay static. D half-lines per wave are kept in flight (D x 4 KiB: memory-level parallelism). The hash
// runs in rolling form over two alternating rings of table values (see scan3_tile); pre-rotated 64x
// replicated table (every lane of a ds_read_b32 hits its own bank whatever the data byte), batched lookups, per-tile slot lists.
// Index algebra restated on CPU: tests/helpers.py::scan_coop_model.
#define PBS_LOOKUP3(w, k) \
    (*reinterpret_cast<const uint32_t *>(lds0 + __builtin_amdgcn_perm((w), lane4, 0x0c0c0400u | ((uint32_t)(k) << 8))))

// one tile of k_scan3. INTERIOR = the whole tile and its 64-byte warm-up lie inside the buffer (wave-uniform by
// construction): the check-free instantiation
template <int LINES, int D, bool INTERIOR>
__device__ __forceinline__ void scan3_tile(const ScanParams &p, const uint64_t t_idx, const uint64_t wbase, const uint64_t A,
                                           const int lane, const uint8_t *lds0, uint8_t *stage, uint32_t *wcnt) {
    constexpr uint32_t SL = LINES * 128;
    constexpr int PITCH = 80;
    constexpr int HALVES = LINES * 2;
    const uint64_t sbase = wbase + (uint64_t)lane * SL;
    const uint32_t thr = p.thr;
    const uint32_t lane4 = (uint32_t)lane << 2;
    const int q4 = lane & ~3, ql = lane & 3;

    // quad-cooperative fetch of half-line `hl` (64-byte units from the strip start; -1 = warm-up window):
    // instruction j serves strip q4 + j, this lane brings piece ql of it
    auto gfetch = [&](uint4 (&G)[4], const int hl) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t a = (int64_t)wbase + (int64_t)(q4 + j) * SL + (int64_t)hl * 64 + ql * 16;
            if constexpr (INTERIOR) {
                const uint4 v = *reinterpret_cast<const uint4 *>(p.data_al + a);
                G[j] = make_uint4(v.x, v.y, v.z, v.w);  // component-wise: a whole-struct store keeps G[][] in scratch
            } else {  // branch-free (a branch per load would force vmcnt(0) waits): clamp the address, select zero
                const bool in = a >= 0 && (uint64_t)a < A;
                const uint4 v = *reinterpret_cast<const uint4 *>(p.data_al + (in ? a : 0));
                G[j] = make_uint4(in ? v.x : 0u, in ? v.y : 0u, in ? v.z : 0u, in ? v.w : 0u);
            }
        }
    };
    // LDS transpose: [strip][piece] in, own row out (LDS operations of one wave execute in order)
    auto transpose = [&](const uint4 (&G)[4], uint4 (&X)[4]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4 *>(stage + (q4 + j) * PITCH + ql * 16) = G[j];
        wave_sync();
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = *reinterpret_cast<const uint4 *>(stage + lane * PITCH + k * 16);
        wave_sync();
    };

    // Rolling form (window 64 == two 32-bit turns, so the outgoing term needs no rotation):
    //   h_i = rotl(h_{i-1}, 1) ^ t_i ^ t_{i-64},   t = pre-rotated table value of the byte
    // The table values of the previous 64 bytes live in a register ring; a half-line is exactly one turn of it,
    // so two rings alternate roles per half-line (lookups land directly in the "new" ring, no copies) and a
    // byte costs: 1 v_perm (LDS address) + 1 ds_read + 1 rotate + 1 three-input xor + 1/2 max3.
    uint32_t ring[2][64];
    uint32_t hc = 0;
    uint4 G[D][4], X[4];
    gfetch(G[0], -1);
    transpose(G[0], X);
#pragma unroll
    for (int d = 0; d < D; ++d) {
        gfetch(G[d], d);
        __builtin_amdgcn_sched_barrier(0);  // keep stage order: the loop's vmcnt waits assume oldest stage first
    }
    {   // warm-up over the 64 bytes before the strip (plays half-line -1: fills ring[1])
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w[4] = {X[g].x, X[g].y, X[g].z, X[g].w};
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const uint32_t t = PBS_LOOKUP3(w[k >> 2], k & 3);
                ring[1][g * 16 + k] = t;
                hc = __builtin_rotateleft32(hc, 1) ^ t;
            }
        }
    }
    // one half-line: transpose stage d, (REFILL) put the stage's next half-line in flight, hash 64 bytes per lane
    auto half = [&](const int d, const int hl, auto refill_c) {
        uint32_t(&rn)[64] = ring[d & 1];        // hl and d have the same parity (D and the loop step are even)
        const uint32_t(&ro)[64] = ring[(d & 1) ^ 1];
        transpose(G[d], X);
        if constexpr (decltype(refill_c)::value) gfetch(G[d], hl + D);
        auto issue = [&](const int batch) {
            const uint4 &v = X[batch >> 1];
            const uint32_t w0 = (batch & 1) ? v.z : v.x, w1 = (batch & 1) ? v.w : v.y;
#pragma unroll
            for (int k = 0; k < 4; ++k) rn[batch * 8 + k] = PBS_LOOKUP3(w0, k);
#pragma unroll
            for (int k = 0; k < 4; ++k) rn[batch * 8 + 4 + k] = PBS_LOOKUP3(w1, k);
        };
        issue(0);
        uint32_t h[16];
        uint32_t acc = 0;
#pragma unroll
        for (int batch = 0; batch < 8; ++batch) {
            if (batch < 7) {
                issue(batch + 1);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC87F);  // lgkmcnt(8)
            } else {
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 8; k += 2) {
                const int pos = batch * 8 + k;  // byte within the half-line == ring index
                const uint32_t h0 = __builtin_amdgcn_bitop3_b32(__builtin_rotateleft32(hc, 1), rn[pos], ro[pos], 0x96);
                const uint32_t h1 = __builtin_amdgcn_bitop3_b32(__builtin_rotateleft32(h0, 1), rn[pos + 1], ro[pos + 1], 0x96);
                h[pos & 15] = h0;
                h[(pos + 1) & 15] = h1;
                hc = h1;
                asm("v_max3_u32 %0, %1, %2, %3" : "=v"(acc) : "v"(acc), "v"(h0), "v"(h1));  // the compiler splits half of these
            }
            if (batch & 1) {
                if (__builtin_expect(acc >= thr, 0)) {  // rare: compact (mask + ctz loop), the hot loop has to stay small
                    uint32_t m = 0;
#pragma unroll
                    for (int k = 0; k < 16; ++k) m |= (h[k] >= thr) ? (1u << k) : 0u;
                    while (m) {
                        const uint32_t k = (uint32_t)__builtin_ctz(m);
                        m &= m - 1;
                        const uint64_t ea = sbase + (uint64_t)hl * 64u + (uint32_t)((batch >> 1) * 16) + k + 1u;
                        if (ea >= (uint64_t)p.lead + kWindow && ea <= A) {
                            const uint32_t slot = atomicAdd(wcnt, 1u);
                            if (slot < p.cap) p.tile_slots[t_idx * p.cap + slot] = (uint32_t)(ea - wbase);
                        }
                    }
                }
                acc = 0;
            }
        }
    };
    // steady state: every stage is refilled unconditionally (a conditional refill makes the compiler's vmcnt
    // bookkeeping pessimistic: it then waits for younger stages too); the last D half-lines are peeled
#pragma unroll 1
    for (int hl0 = 0; hl0 < HALVES - D; hl0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) half(d, hl0 + d, std::true_type{});
    }
#pragma unroll
    for (int d = 0; d < D; ++d) half(d, HALVES - D + d, std::false_type{});
}

template <int LINES, int D>
__global__ __launch_bounds__(512, 2) void k_scan3(ScanParams p) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ __attribute__((aligned(1024))) uint32_t tab[256 * 64];  // [entry][lane]
    constexpr uint32_t SL = LINES * 128;
    constexpr uint64_t TILE = 64ull * SL;
    constexpr int PITCH = 80;
    static_assert((LINES * 2) % D == 0 && D % 2 == 0, "the half-line loop is unrolled by the (even) prefetch depth");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t *counters = reinterpret_cast<uint32_t *>(smem);
    uint8_t *stage = smem + 64 + wave * (64 * PITCH);
    for (int i = tid; i < 256 * 64; i += 512) tab[i] = p.table_rot[i >> 6];
    __syncthreads();

    uint32_t *wcnt = counters + wave;
    const uint64_t A = p.nbytes + p.lead;
    const uint8_t *lds0 = reinterpret_cast<const uint8_t *>(tab);

    // tiles_per_wave > 0: the workgroup retires after that many tiles per wave, so that the chip's dispatcher can place
    // other kernels' workgroups (the small resolve-chain kernels and SHA launches of the other batches in flight) on the
    // CU every fraction of a millisecond instead of after the whole scan (see launch_scan3)
    for (uint32_t done = 0; p.tiles_per_wave == 0 || done < p.tiles_per_wave; ++done) {
        unsigned long long g0 = 0;
        if (lane == 0) g0 = atomicAdd(p.tile_queue, 1ull);
        const uint64_t t_idx = __shfl(g0, 0, 64);
        if (t_idx >= p.ntiles) break;
        // flat range: tile t starts at t * TILE. Page ring: the tile's page comes from the round's page table — the
        // page's body is preceded by a 128-byte pad that holds the last bytes of the stream's previous page (the
        // window warm-up of the page's first strip), and only its `valid` bytes take part
        uint64_t wbase = t_idx * TILE, At = A;
        if (p.ring_pages) {
            const RingPage &pe = p.ring_pages[t_idx / p.ring_tpp];
            wbase = pe.phys_off + (t_idx % p.ring_tpp) * TILE;
            At = pe.phys_off + pe.valid;
        }
        if (lane == 0) *wcnt = 0;
        wave_sync();
        if (wbase < At) {
            if ((wbase >= 64) && (wbase + TILE <= At))
                scan3_tile<LINES, D, true>(p, t_idx, wbase, At, lane, lds0, stage, wcnt);
            else
                scan3_tile<LINES, D, false>(p, t_idx, wbase, At, lane, lds0, stage, wcnt);
        }
        wave_sync();
        if (lane == 0) p.tile_cnt[t_idx] = *wcnt;
        wave_sync();
    }
}
#undef PBS_LOOKUP3

template <int LINES, int D>
static hipError_t launch_scan3(const ScanParams &p, int num_cus, hipStream_t st) {
    constexpr size_t lds = 64 + 8 * 64 * 80;  // counters + 8 per-wave stages (40 KiB: also keeps SHA workgroups off this CU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_scan3<LINES, D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    uint64_t blocks = (p.ntiles + 7) / 8;
    // A scan workgroup fills its CU completely (244 VGPRs x 2 waves per SIMD, 104 KiB LDS). With PERSISTENT workgroups
    // (one per CU draining the tile queue) no other kernel can place a single wave anywhere for the 15-25 ms of a 64 GiB
    // scan — measured with 4 batches in flight: the small resolve-chain kernels of the OTHER batches took 11 ms instead
    // of 0.4-4.9 and their SHA launches started ~24 ms late. Hence a few CUs are never taken (16 of 256). (Workgroups that
    // retire after a few tiles so that the dispatcher interleaves everyone else were measured too: the resolve chains drop
    // from 13 to 9 ms but the scans stretch from 25 to 40 ms, no net gain — ScanParams::tiles_per_wave stays 0.)
    const uint64_t usable = (uint64_t)std::max(1, num_cus - (num_cus >= 64 ? 16 : 0));
    if (blocks > usable) blocks = usable;
    if (p.max_blocks && blocks > p.max_blocks) blocks = p.max_blocks;
    hipLaunchKernelGGL((k_scan3<LINES, D>), dim3((unsigned)blocks), dim3(512), lds, st, p);
    return hipGetLastError();
}

// half-lines in flight per wave: 4 (at 2 the kernel measured 4.66 instead of 4.90 TB/s, at 1 it is latency-bound: 3.96)
template <int LINES>
static hipError_t launch_scan3_any(const ScanParams &p, int num_cus, hipStream_t st) {
    return launch_scan3<LINES, 4>(p, num_cus, st);
}

uint32_t scan_tile_bytes(uint64_t nbytes) {
    return nbytes < (48ull << 20) ? 64u * 4u * 128u : 64u * 34u * 128u;  // (small batches: small tiles, so that the chip fills)
}

hipError_t launch_scan(const ScanParams &p, int num_cus, hipStream_t st) {
    if (p.ntiles == 0) return hipSuccess;
    if (p.tile_bytes == 64u * 34u * 128u) return launch_scan3_any<34>(p, num_cus, st);
    if (p.tile_bytes == 64u * 4u * 128u) return launch_scan3_any<4>(p, num_cus, st);
    return hipErrorInvalidValue;
}

// =====================================================================================
// exclusive scan of uint32 counts (three small kernels; inputs are <= a few M entries)
// =====================================================================================
constexpr int kScanBlockElems = 1024;  // 256 threads x 4

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t n = __shfl_up(v, d, 64);
        if (lane >= d) v += n;
    }
    return v;
}

// block-wide exclusive scan of one value per thread (256 threads); returns exclusive prefix,
// *block_total gets the sum
__device__ __forceinline__ uint32_t block_excl_scan_256(uint32_t v, uint32_t *block_total) {
    __shared__ uint32_t wsum[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    *block_total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return base + incl - v;
}

__global__ __launch_bounds__(256) void k_scan_sums(const uint32_t *in, uint64_t n, uint32_t clamp, uint32_t *bsum,
                                                   uint32_t *maxval) {
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlockElems + (uint64_t)threadIdx.x * 4;
    uint32_t s = 0, m = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) {
            const uint32_t v = in[base + k];
            m = max(m, v);
            s += min(v, clamp);
        }
    }
    uint32_t total;
    (void)block_excl_scan_256(s, &total);
    // block max via wave reduce + atomic (rarely contended: one atomic per wave, only if it raises the max)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor(m, d, 64));
    if ((threadIdx.x & 63) == 0 && maxval && m > 0) atomicMax(maxval, m);
    if (threadIdx.x == 0) bsum[blockIdx.x] = total;
}

// single block: in-place exclusive scan of bsum[0..nb), total -> *total
__global__ __launch_bounds__(256) void k_scan_bsums(uint32_t *bsum, uint64_t nb, uint32_t *total) {
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint64_t c = 0; c < nb; c += 256) {
        const uint64_t i = c + threadIdx.x;
        const uint32_t v = (i < nb) ? bsum[i] : 0u;
        uint32_t tot;
        const uint32_t ex = block_excl_scan_256(v, &tot);
        const uint32_t carry = carry_s;
        if (i < nb) bsum[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = carry_s;
}

__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t *in, uint64_t n, uint32_t clamp,
                                                    const uint32_t *bsum, uint32_t *out) {
    const uint64_t base = (uint64_t)blockIdx.x * kScanBlockElems + (uint64_t)threadIdx.x * 4;
    uint32_t v[4];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = (base + k < n) ? min(in[base + k], clamp) : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t ex = block_excl_scan_256(s, &total) + bsum[blockIdx.x];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (base + k < n) out[base + k] = ex;
        ex += v[k];
    }
}

size_t scan_tmp_words(uint64_t n) { return (size_t)((n + kScanBlockElems - 1) / kScanBlockElems) + 8; }

hipError_t launch_exclusive_scan(const uint32_t *in, uint64_t n, uint32_t clamp, uint32_t *out, uint32_t *total,
                                 uint32_t *maxval, uint32_t *tmp, hipStream_t st) {
    if (n == 0) return hipMemsetAsync(total, 0, sizeof(uint32_t), st);
    const uint64_t nb = (n + kScanBlockElems - 1) / kScanBlockElems;
    hipLaunchKernelGGL(k_scan_sums, dim3((unsigned)nb), dim3(256), 0, st, in, n, clamp, tmp, maxval);
    hipLaunchKernelGGL(k_scan_bsums, dim3(1), dim3(256), 0, st, tmp, nb, total);
    hipLaunchKernelGGL(k_scan_apply, dim3((unsigned)nb), dim3(256), 0, st, in, n, clamp, tmp, out);
    return hipGetLastError();
}

// =====================================================================================
// (2a) compaction: per-tile unordered slots -> dense ascending END offsets (caller coords)
// =====================================================================================
__global__ __launch_bounds__(256) void k_compact(const uint32_t *tile_cnt, const uint32_t *tile_off,
                                                 const uint32_t *tile_slots, uint32_t cap, uint64_t ntiles,
                                                 uint32_t lead, uint64_t *dense, uint64_t dense_cap,
                                                 uint32_t tile_bytes) {
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntiles) return;
    const uint32_t c = min(tile_cnt[t], cap);  // (a tile that found more keeps the first `cap` it recorded: see DenseTiles)
    if (c == 0) return;
    const uint32_t *sl = tile_slots + t * cap;
    const uint64_t base = tile_off[t];
    const uint64_t tbase = t * (uint64_t)tile_bytes;
    if (c <= 48) {  // the normal case (a handful of candidates per tile): rank by comparison
        for (uint32_t j = 0; j < c; ++j) {
            const uint32_t vj = sl[j];
            uint32_t rank = 0;
            for (uint32_t i = 0; i < c; ++i) rank += (sl[i] < vj) ? 1u : 0u;  // offsets within a tile are distinct
            if (base + rank < dense_cap) dense[base + rank] = tbase + vj - lead;
        }
        return;
    }
    // Dense tile (periodic / crafted input; the capacity-retry path): linear-time stable counting sort by lane strip.
    // A lane appends its own hits in ascending order (it walks its strip forwards and its LDS atomics execute in
    // program order), so the slot list is ascending WITHIN each strip and only interleaved ACROSS strips.
    const uint32_t strip = tile_bytes / 64u;
    uint32_t cnt[64];
    for (int i = 0; i < 64; ++i) cnt[i] = 0;
    for (uint32_t j = 0; j < c; ++j) cnt[min((sl[j] - 1u) / strip, 63u)]++;  // END offsets: 1..tile_bytes
    uint32_t run = 0;
    for (int i = 0; i < 64; ++i) {
        const uint32_t k = cnt[i];
        cnt[i] = run;
        run += k;
    }
    for (uint32_t j = 0; j < c; ++j) {
        const uint32_t vj = sl[j];
        const uint32_t pos = cnt[min((vj - 1u) / strip, 63u)]++;
        if (base + pos < dense_cap) dense[base + pos] = tbase + vj - lead;
    }
}

hipError_t launch_compact(const uint32_t *tile_cnt, const uint32_t *tile_off, const uint32_t *tile_slots,
                          uint32_t cap, uint64_t ntiles, uint32_t lead, uint64_t nbytes, uint64_t *dense,
                          uint64_t dense_cap, uint32_t tile_bytes, hipStream_t st) {
    (void)nbytes;
    if (ntiles == 0) return hipSuccess;
    const uint64_t nb = (ntiles + 255) / 256;
    hipLaunchKernelGGL(k_compact, dim3((unsigned)nb), dim3(256), 0, st, tile_cnt, tile_off, tile_slots, cap, ntiles,
                       lead, dense, dense_cap, tile_bytes);
    return hipGetLastError();
}

// =====================================================================================
// (2b) min/max resolution — one wave per segment walks the sorted candidate list
// =====================================================================================
// Serial rule being reproduced (oracle/buzhash_oracle.c shall_break): after a cut at s the
// next cut is the first end e with (e - s >= max) or (e - s >= max(min, 65) and e is a
// candidate); the stream end closes the last chunk. The break test only runs in the
// rolling loop, i.e. from chunk_size 65 on, hence the 65.
// Suggested boundaries (payload chunker, SURVEY.md Appendix A note 3 / E.3; oracle_payload_chunker_scan fed byte by
// byte): a second ascending list per segment. After a cut at s the next cut is the EARLIER of the hash/max cut and
// the first suggested boundary b with min <= b - s <= max; boundaries with b - s < min are dropped for good. For
// min >= 65 they behave exactly like extra candidates; the separate list keeps the min = 64 corner (hash cuts need
// chunk_size >= 65, a suggested one only >= min) exact.
// ---- candidate-dense tiles: the first TRUE candidate of a stretch of one tile, on demand (see DenseTiles, kernels.h) ----
// All 64 lanes call it with the same arguments. q = the byte at the END of the first window (the window of position i is
// q[i - 63 .. i]); n positions. Returns the index of the first position whose window hash passes the break test, ~0u if none.
// One wave, rows of 256 bytes (one aligned dword per lane, coalesced), the prefix form the scan kernels use:
//   Q(x) = rotl(Q(x - 1), 1) ^ t(x)   (t = pre-rotated table value of byte x),   h(x) = Q(x) ^ Q(x - 64)   (64 = 0 mod 32)
// Q inside a lane's 4 bytes is 3 rotate-xor steps; across lanes an inclusive scan with a rotation of 4 bits per lane (six
// shuffle steps: 4, 8, 16, then 32 / 64 / 128 = 0 bits); Q(x - 64) is the same byte of the lane 16 below. Lanes 0-15 of a
// row only warm the prefix up, so rows advance by 192 bytes. A dword that holds no byte of [first window start, last
// window end] is never read (the caller's buffer may begin or end right there). ~90 instructions and 11 shuffles per 192
// positions, a dozen VGPRs: this is the cold path of the resolve walk and must not cost the walk its registers.
__device__ __forceinline__ uint32_t dense_first_hit(const uint8_t *q, const uint32_t n, const uint32_t *__restrict__ tab,
                                                    const uint32_t thr) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint8_t *w0 = q - 63;                                        // first byte of the first window
    const uint32_t sh = (uint32_t)((uintptr_t)w0 & 3u);
    const uint8_t *a0 = w0 - sh - 4;                                   // row 0 starts one dword early: position 0 ends at byte
                                                                       // sh + 67 >= 64 of its row, i.e. in a lane >= 16
    const uintptr_t lo_dw = (uintptr_t)(w0 - sh), hi_dw = (uintptr_t)(q + (n - 1u)) & ~(uintptr_t)3;  // first / last dword with a needed byte
    auto load_row = [&](const uint32_t r) -> uint32_t {
        const uintptr_t A = (uintptr_t)a0 + 192u * (uintptr_t)r + 4u * lane;
        const uintptr_t Ac = min(max(A, lo_dw), hi_dw);
        const uint32_t v = *reinterpret_cast<const uint32_t *>(Ac);
        return (A == Ac) ? v : 0u;
    };
    const uint32_t rows = (n + sh + 3u) / 192u + 1u;                   // positions of row r: 192 r + 4 lane + k - sh - 67
    uint32_t w = load_row(0);
    for (uint32_t r = 0; r < rows; ++r) {
        const uint32_t wn = (r + 1u < rows) ? load_row(r + 1u) : 0u;   // next row's bytes in flight behind this row's arithmetic
        const uint32_t t0 = tab[w & 0xffu], t1 = tab[(w >> 8) & 0xffu], t2 = tab[(w >> 16) & 0xffu], t3 = tab[w >> 24];
        const uint32_t q0 = t0;
        const uint32_t q1 = __builtin_rotateleft32(q0, 1) ^ t1;
        const uint32_t q2 = __builtin_rotateleft32(q1, 1) ^ t2;
        const uint32_t q3 = __builtin_rotateleft32(q2, 1) ^ t3;
        uint32_t X = q3;                                               // inclusive scan over the lanes' 4-byte sums
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(X, d, 64);
            if ((int)lane >= d) X ^= __builtin_rotateleft32(y, (4 * d) & 31);
        }
        uint32_t Xp = __shfl_up(X, 1, 64);
        if (lane == 0) Xp = 0;
        const uint32_t Q[4] = {q0 ^ __builtin_rotateleft32(Xp, 1), q1 ^ __builtin_rotateleft32(Xp, 2),
                               q2 ^ __builtin_rotateleft32(Xp, 3), q3 ^ __builtin_rotateleft32(Xp, 4)};
        uint32_t first = ~0u;
        const uint32_t i0 = 192u * r + 4u * lane - sh - 67u;          // position of byte 0 of this lane (wraps below 0: masked)
#pragma unroll
        for (int k = 3; k >= 0; --k) {
            const uint32_t h = Q[k] ^ __shfl_up(Q[k], 16, 64);
            const uint32_t i = i0 + (uint32_t)k;
            if (lane >= 16u && h >= thr && i < n) first = i;           // (i < n also rejects the wrapped "negative" positions)
        }
        const unsigned long long m = __ballot(first != ~0u);
        if (m) return __shfl(first, __ffsll((long long)m) - 1, 64);    // lanes are in position order
        w = wn;
    }
    return ~0u;
}

// The walk is about to take `c` (first LISTED candidate >= tlo, ~0 = none) as the next hash cut unless the maximum (`lim` =
// min(s + max, B)) comes first. Tiles that overflowed their slots may hold an unlisted candidate before that: the first
// one in [tlo, min(c - 1, lim)] is returned instead of c. Wave-uniform.
__device__ __forceinline__ uint64_t dense_refine(const DenseTiles &d, const DenseSeg &ds, const uint64_t tlo, const uint64_t lim,
                                                 const uint64_t c) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lo = max(tlo, ds.L0 + 1u);  // (page ring: older pages hold no candidate at or behind tlo)
    const uint64_t hi = min(c - 1u, lim);
    if (ds.ntiles == 0 || lo > hi) return c;
    // tile coordinates x = E - L0 + lead (>= 1): tile k holds the END offsets x in (k * tile_bytes, (k + 1) * tile_bytes]
    const uint64_t lo_x = lo - ds.L0 + ds.lead, hi_x = min(hi - ds.L0 + ds.lead, ds.ntiles * d.tile_bytes);
    if (lo_x > hi_x) return c;
    const uint64_t ta = (lo_x - 1u) / d.tile_bytes, tb = (hi_x - 1u) / d.tile_bytes;
    for (uint64_t t0 = ta; t0 <= tb; t0 += 64u) {
        const uint64_t t = t0 + lane;
        unsigned long long m = __ballot(t <= tb && d.tile_cnt[ds.first_tile + t] > d.cap);
        while (m) {
            const uint64_t tt = t0 + (uint64_t)(__ffsll((long long)m) - 1);
            m &= m - 1ull;
            const uint64_t qlo_x = max(lo_x, tt * d.tile_bytes + 1u), qhi_x = min(hi_x, (tt + 1u) * d.tile_bytes);
            const uint64_t E0 = ds.L0 + qlo_x - ds.lead;  // the stretch's first END offset, in the walk's coordinates
            const uint8_t *q;
            if (d.pages) {
                const RingPage &pe = d.pages[(ds.first_tile + tt) / d.tpp];
                q = d.base + pe.phys_off + (E0 - 1u - pe.logical);
            } else {
                q = d.base + (E0 - 1u);
            }
            const uint32_t r = dense_first_hit(q, (uint32_t)(qhi_x - qlo_x + 1u), d.table_rot, d.thr);
            if (r != ~0u) return E0 + r;
        }
    }
    return c;
}

// The walk of ONE segment by ONE wave (all 64 lanes call it with the same arguments). Returns the number of records;
// WRITE: record k goes to recs[rbase + k] (lane 0), `seg` is stored in its segment field.
// Page-ring rounds (`ring`; ring_kernels.inc): the segment is a stream's open chunk + its new pages in logical
// coordinates ((slot << kRingOffBits) | offset); suggested offsets are relative to the STREAM's byte 0; the segment's end is the
// stream's end only if RingSeg::final; and the walk reports what the round leaves behind: is the last record the still-open
// chunk (it is unless the serial chunker cuts exactly at the current end: a max-size chunk, a candidate or a winning
// suggested boundary there), where that chunk starts, and — reader-buffer rule only — the hash candidate inside it that a
// boundary beyond the bytes seen so far pre-empts. Such a candidate lies in pages the next round does not scan again, so
// it is carried in the stream state and handed back as `ecand_first`.
template <bool WRITE>
__device__ __forceinline__ uint32_t resolve_walk(const uint64_t *cands, const uint64_t n, const uint64_t A, const uint64_t B,
                                                 const uint32_t seg, const uint32_t effmin, const uint32_t maxsz,
                                                 const uint64_t rbase, pbsgpu_record *recs, const uint64_t rec_cap,
                                                 const uint64_t *sugg, const uint64_t sbeg, const uint64_t send,
                                                 const uint32_t cmin, const SuggFeed fr, const bool ring,
                                                 const uint64_t ecand_first, bool *last_real_out, uint64_t *last_start_out,
                                                 uint64_t *ecand_out, const DenseTiles &dz, const DenseSeg &ds) {
    const int lane = threadIdx.x & 63;
    uint64_t s = A;
    uint32_t k = 0;
    // WRITE: record k waits in the registers of lane k mod 64 and 64 records leave with ONE store instruction. (Until round 6 lane 0
    // stored every record as it was cut — and the loop's header waits for vmcnt(0), a loaded window may be pending there, which
    // on this ISA also counts the stores: every record paid a store round trip, ~1.2 us beside the ring's streaming traffic;
    // 87 of the control kernel's 147 us per round: profiles/r06_control_kernel_phases.log.)
    [[maybe_unused]] uint64_t my_end = 0;
    [[maybe_unused]] uint32_t my_size = 0;
    // coordinates of the suggested offsets / of the absolute reader grid: the segment start, or (ring) the stream's byte 0
    const uint64_t P = ring ? (A & ~kRingOffMask) : A;
    bool last_real = true;
    uint64_t last_start = A, last_ecand = ~0ull;

    // first candidate index with value >= A + effmin (uniform binary search)
    uint64_t lo = 0, hi = n;
    const uint64_t first = A + effmin;
    while (lo < hi) {
        const uint64_t mid = (lo + hi) >> 1;
        if (cands[mid] < first) lo = mid + 1; else hi = mid;
    }
    uint64_t wb = lo;  // window base: lane holds cands[wb + lane]
    uint64_t cv = (wb + lane < n) ? cands[wb + lane] : ~0ull;
    // suggested boundaries of this segment: same wave-wide window walk
    uint64_t swb = sbeg;
    uint64_t sv = (swb + lane < send) ? P + sugg[swb + lane] : ~0ull;

    while (s < B) {
        const uint64_t tlo = s + effmin, thi = s + maxsz;
        uint64_t c;
        for (;;) {
            const unsigned long long m = __ballot(cv >= tlo);
            if (m) {
                const int j = __ffsll((long long)m) - 1;
                c = __shfl(cv, j, 64);
                break;
            }
            wb += 64;
            // none of these 64 and none of the next 64 either: candidate-dense data lists thousands of candidates per MiB (a
            // subset of an overflowed tile's: DenseTiles) — jump to the first one >= tlo by binary search instead of stepping
            // (a stream whose every position is a candidate: 41 us per cut stepping through ~8 k entries, measured)
            if (wb + 63 < n && cands[wb + 63] < tlo) {
                uint64_t l2 = wb + 64, h2 = n;
                while (l2 < h2) {
                    const uint64_t mid = (l2 + h2) >> 1;
                    if (cands[mid] < tlo) l2 = mid + 1; else h2 = mid;
                }
                wb = l2;
            }
            cv = (wb + lane < n) ? cands[wb + lane] : ~0ull;
        }
        if (__builtin_expect(dz.tile_cnt != nullptr, 0)) c = dense_refine(dz, ds, tlo, min(thi, B), c);  // (some tile overflowed its slots)
        if (s == A && ecand_first != ~0ull) c = ecand_first;  // older than every listed candidate, >= tlo by construction
        const uint64_t e0 = (c < thi) ? c : thi;  // where the hash / max rule cuts, were the bytes there
        uint64_t e = (e0 > B) ? B : e0;
        bool real = e0 <= B;
        uint64_t ec = ~0ull;
        if (send > sbeg) {  // segment-uniform
            const uint64_t slo = s + cmin;
            uint64_t b;
            for (;;) {
                const unsigned long long m = __ballot(sv >= slo);  // lanes past the list hold ~0: always terminates
                if (m) {
                    b = __shfl(sv, __ffsll((long long)m) - 1, 64);
                    break;
                }
                swb += 64;
                sv = (swb + lane < send) ? P + sugg[swb + lane] : ~0ull;
            }
            if (fr.feed <= 1) {
                if (b <= e) {  // b >= s + min and b <= e <= s + max: a legal chunk (b == e: the same position, and a cut)
                    e = b;
                    real = true;
                }
            } else if (b <= B && b - s <= maxsz) {
                // the reference's payload chunker sees `feed` bytes per scan call: a boundary inside the call's buffer is
                // taken before the hash scan of that buffer runs, so it also wins over an EARLIER hash cut in the same
                // buffer; buffers are counted from the last cut, or (absolute) from the stream start
                const uint64_t ob = fr.absolute ? fr.origin + (b - P) : b - s, oe = fr.absolute ? fr.origin + (e - P) : e - s;
                const uint64_t jb = (ob - 1) / fr.feed, je = (oe - 1) / fr.feed;
                if (jb <= je) {
                    e = b;
                    real = true;
                }
            } else if (fr.open_end && b != ~0ull && b > B && b - s <= maxsz) {
                const uint64_t ob = fr.absolute ? fr.origin + (b - P) : b - s, oe = fr.absolute ? fr.origin + (e - P) : e - s;
                if ((ob - 1) / fr.feed <= (oe - 1) / fr.feed) {  // the cut is at b, beyond the bytes that are here: open chunk
                    if (e0 <= B && e0 == c) ec = c;              // ... and the hash cut it pre-empts still counts if the stream ends first
                    e = B;
                    real = false;
                }
            }
        }
        if (WRITE) {
            if (lane == (int)(k & 63u)) {
                my_end = e - A;
                my_size = (uint32_t)(e - s);
            }
            if ((k & 63u) == 63u) {  // records k - 63 .. k
                const uint64_t idx = rbase + (uint64_t)(k - 63u) + (uint64_t)lane;
                if (idx < rec_cap) {
                    pbsgpu_record *r = recs + idx;
                    r->end = my_end;
                    r->segment = seg;
                    r->size = my_size;
                }
            }
        }
        last_real = real;
        last_start = s;
        last_ecand = ec;
        ++k;
        s = e;
    }
    if (WRITE) {  // the last, partial group
        const uint32_t rest = k & 63u;
        const uint64_t idx = rbase + (uint64_t)(k - rest) + (uint64_t)lane;
        if ((uint32_t)lane < rest && idx < rec_cap) {
            pbsgpu_record *r = recs + idx;
            r->end = my_end;
            r->segment = seg;
            r->size = my_size;
        }
    }
    *last_real_out = last_real;
    *last_start_out = last_start;
    *ecand_out = last_ecand;
    return k;
}

// (`dz` non-null tile_cnt: the batch was scanned at the capacity LIMIT and `*maxcnt_p` tells whether any tile overflowed it)
template <bool WRITE>
__global__ __launch_bounds__(256) void k_resolve(const uint64_t *cands, const uint32_t *ncand_p,
                                                 const pbsgpu_segment *segs, uint32_t nseg, uint32_t effmin,
                                                 uint32_t maxsz, uint32_t *seg_cnt, const uint32_t *seg_off,
                                                 pbsgpu_record *recs, uint64_t rec_cap, const uint64_t *sugg,
                                                 const uint32_t *sugg_idx, uint32_t cmin, const uint32_t *gate,
                                                 SuggFeed fr, DenseTiles dz, DenseSeg ds) {
    const uint32_t seg = (uint32_t)(((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    const int lane = threadIdx.x & 63;
    if (seg >= nseg) return;
    if (gate && *gate == 0) return;  // fallback launch behind k_resolve_par: only if that kernel handed the job back
    __builtin_amdgcn_s_setprio(2);  // a latency-bound serial walk: do not queue behind throughput waves on this SIMD
    const uint64_t n = *ncand_p;
    const uint64_t A = segs[seg].offset, B = A + segs[seg].length;
    bool last_real;
    uint64_t last_start, ecand;
    const uint32_t k = resolve_walk<WRITE>(cands, n, A, B, seg, effmin, maxsz, WRITE ? (uint64_t)seg_off[seg] : 0, recs, rec_cap,
                                           sugg, sugg ? sugg_idx[seg] : 0, sugg ? sugg_idx[seg + 1] : 0, cmin, fr, false, ~0ull,
                                           &last_real, &last_start, &ecand, dz, ds);
    if (lane == 0 && seg_cnt) seg_cnt[seg] = k;  // count pass; also the single-segment write pass (count -> *nrec)
}

static DenseSeg flat_dense_seg(const DenseTiles *dz, uint32_t lead, uint64_t ntiles) {
    DenseSeg ds{};
    if (dz) {
        ds.lead = lead;
        ds.ntiles = ntiles;
    }
    return ds;
}

hipError_t launch_resolve_count(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                uint32_t nseg, uint32_t effmin, uint32_t maxsz, uint32_t *seg_cnt, const Suggested &sg,
                                hipStream_t st, const DenseTiles *dz, uint32_t lead, uint64_t ntiles) {
    if (nseg == 0) return hipSuccess;
    const uint64_t nb = ((uint64_t)nseg * 64 + 255) / 256;
    hipLaunchKernelGGL((k_resolve<false>), dim3((unsigned)nb), dim3(256), 0, st, cands, ncand, segs, nseg, effmin,
                       maxsz, seg_cnt, (const uint32_t *)nullptr, (pbsgpu_record *)nullptr, (uint64_t)0, sg.offsets,
                       sg.index, sg.cmin, (const uint32_t *)nullptr, SuggFeed{sg.feed, sg.origin, sg.absolute, sg.open_end},
                       dz ? *dz : DenseTiles{}, flat_dense_seg(dz, lead, ntiles));
    return hipGetLastError();
}

hipError_t launch_resolve_write(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                uint32_t nseg, uint32_t effmin, uint32_t maxsz, const uint32_t *seg_off,
                                pbsgpu_record *recs, uint64_t rec_cap, const Suggested &sg, hipStream_t st,
                                const DenseTiles *dz, uint32_t lead, uint64_t ntiles) {
    if (nseg == 0) return hipSuccess;
    const uint64_t nb = ((uint64_t)nseg * 64 + 255) / 256;
    hipLaunchKernelGGL((k_resolve<true>), dim3((unsigned)nb), dim3(256), 0, st, cands, ncand, segs, nseg, effmin,
                       maxsz, (uint32_t *)nullptr, seg_off, recs, rec_cap, sg.offsets, sg.index, sg.cmin,
                       (const uint32_t *)nullptr, SuggFeed{sg.feed, sg.origin, sg.absolute, sg.open_end},
                       dz ? *dz : DenseTiles{}, flat_dense_seg(dz, lead, ntiles));
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------
// One long stream, resolved in parallel. The serial walk above costs ~0.26 us per chunk on a single wave (4.6 ms for
// the 17.7 k chunks of a 64 GiB stream, 11 ms when that wave shares its SIMD with SHA waves). The cut chain is a
// FUNCTION on nodes {stream start, every candidate}: next(v) = the first candidate cut reached from a cut at v (possibly
// after some forced max-size cuts), so it can be followed by pointer doubling:
//   1. every node computes next(v), the number of records of that hop (forced cuts + the closing one) and its end;
//   2. doubling tables J_m = next^(2^m) and R_m = records along 2^m hops (log2(n) rounds, one workgroup);
//   3. the k-th hop of the path from the stream start is found independently for every k by decomposing k in binary,
//      and writes its records at the prefix count R gives it.
// ~20 barrier-separated rounds of a few memory accesses each instead of 17 k dependent iterations. Falls back to the
// serial walk when the node count exceeds the scratch capacity (dense / crafted inputs) — same kernel, wave 0.
constexpr uint32_t kParEnd = 0xffffffffu;  // terminal node id

__global__ __launch_bounds__(1024) void k_resolve_par(const uint64_t *cands, const uint32_t *ncand_p, const pbsgpu_segment *segs,
                                                      uint32_t effmin, uint32_t maxsz, uint32_t *nrec_out, pbsgpu_record *recs,
                                                      uint64_t rec_cap, uint32_t *J, uint32_t *R, uint64_t *endpos,
                                                      uint32_t node_cap, uint32_t levels_cap, uint32_t *fallback) {
    const uint64_t n64 = *ncand_p;
    const uint64_t A = segs[0].offset, B = A + segs[0].length;
    const uint32_t tid = threadIdx.x, nth = blockDim.x;
    // nodes: 0 = stream start (position A), i + 1 = candidate i
    const uint64_t nodes64 = n64 + 1;
    uint32_t levels = 1;
    while ((1ull << levels) < nodes64 + 1) ++levels;
    if (nodes64 > node_cap || levels > levels_cap || B == A) {  // uniform: let the serial kernel do it
        if (tid == 0) *fallback = (B == A) ? 0u : 1u;
        if (B == A && tid == 0) *nrec_out = 0;
        return;
    }
    if (tid == 0) *fallback = 0;
    const uint32_t nodes = (uint32_t)nodes64, n = (uint32_t)n64;
    // ---- 1. next(v) for every node
    for (uint32_t v = tid; v < nodes; v += nth) {
        uint64_t s = (v == 0) ? A : cands[v - 1];
        uint32_t k = 0, nx = kParEnd;
        uint64_t e = B;
        if (s >= B || (v && s <= A)) {  // a candidate at/after the end or before the start never becomes a cut
            J[v] = kParEnd;
            R[v] = 0;
            endpos[v] = B;
            continue;
        }
        uint32_t lo = v;  // candidates are ascending: the next cut lies behind this node's own candidate
        for (;;) {
            const uint64_t tlo = s + effmin, thi = s + maxsz;
            uint32_t hi = n;  // first candidate index j >= lo with cands[j] >= tlo
            while (lo < hi) {
                const uint32_t mid = (lo + hi) >> 1;
                if (cands[mid] < tlo) lo = mid + 1; else hi = mid;
            }
            const uint64_t c = (lo < n) ? cands[lo] : ~0ull;
            if (c < thi && c <= B) {  // a candidate closes the chunk
                e = c;
                nx = (c < B) ? lo + 1 : kParEnd;
                ++k;
                break;
            }
            if (thi >= B) {  // the stream end closes it
                e = B;
                nx = kParEnd;
                ++k;
                break;
            }
            if (lo >= n) {  // no candidate left at all: forced cuts to the end, in one step
                const uint64_t more = (B - s + maxsz - 1) / maxsz;  // chunks from s to B
                k += (uint32_t)more;
                e = B;
                nx = kParEnd;
                break;
            }
            s = thi;  // forced cut at max size, keep walking
            ++k;
        }
        J[v] = nx;
        R[v] = k;
        endpos[v] = e;
    }
    __syncthreads();
    // ---- 2. doubling tables (level m at offset m * node_cap)
    for (uint32_t m = 1; m < levels; ++m) {
        const uint32_t *Jp = J + (size_t)(m - 1) * node_cap, *Rp = R + (size_t)(m - 1) * node_cap;
        uint32_t *Jm = J + (size_t)m * node_cap, *Rm = R + (size_t)m * node_cap;
        for (uint32_t v = tid; v < nodes; v += nth) {
            const uint32_t a = Jp[v];
            if (a == kParEnd) {
                Jm[v] = kParEnd;
                Rm[v] = Rp[v];
            } else {
                Jm[v] = Jp[a];
                Rm[v] = Rp[v] + Rp[a];
            }
        }
        __syncthreads();
    }
    // ---- 3. hops on the path from the start, and the total record count
    __shared__ uint32_t s_hops, s_total;
    if (tid == 0) {
        uint32_t pos = 0, hops = 0, total = 0;
        for (int m = (int)levels - 1; m >= 0; --m) {
            const uint32_t a = J[(size_t)m * node_cap + pos];
            if (a != kParEnd) {
                total += R[(size_t)m * node_cap + pos];
                hops += 1u << m;
                pos = a;
            }
        }
        s_hops = hops + 1;               // the last hop ends in the terminal node
        s_total = total + R[pos];
        *nrec_out = s_total;
    }
    __syncthreads();
    const uint32_t hops = s_hops;
    for (uint32_t h = tid; h < hops; h += nth) {
        uint32_t pos = 0, off = 0;       // node of hop h and records written before it
        for (int m = (int)levels - 1; m >= 0; --m)
            if (h & (1u << m)) {
                off += R[(size_t)m * node_cap + pos];
                pos = J[(size_t)m * node_cap + pos];
            }
        uint64_t s = (pos == 0) ? A : cands[pos - 1];
        const uint32_t k = R[pos];
        const uint64_t e = endpos[pos];
        for (uint32_t t = 0; t < k; ++t) {  // k - 1 forced max-size cuts, then the closing cut at e
            const uint64_t end = (t + 1 < k) ? s + maxsz : e;
            if ((uint64_t)off + t < rec_cap) {
                pbsgpu_record *r = recs + off + t;
                r->end = end - A;
                r->segment = 0;
                r->size = (uint32_t)(end - s);
            }
            s = end;
        }
    }
}

// ---- the same algorithm for MANY candidates (small average chunk sizes: a 64 GiB stream at avg 64 KiB has 1.5 M
// candidates and 1.1 M chunks — the serial walk takes 278 ms there, the single workgroup above would crawl): one
// grid-wide launch per phase / doubling level instead of barriers inside one workgroup.
__device__ __forceinline__ bool par_active(const uint32_t *ncand_p, uint32_t node_cap) {
    return (uint64_t)*ncand_p + 1 <= node_cap;
}

__global__ __launch_bounds__(256) void k_par_next(const uint64_t *cands, const uint32_t *ncand_p, const pbsgpu_segment *segs,
                                                  uint32_t effmin, uint32_t maxsz, uint32_t *J, uint32_t *R, uint64_t *endpos,
                                                  uint32_t node_cap, uint32_t *fallback, uint32_t *nrec_out) {
    const uint64_t A = segs[0].offset, B = A + segs[0].length;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    const bool ok = par_active(ncand_p, node_cap) && B > A;
    if (v == 0) {
        *fallback = (ok || B == A) ? 0u : 1u;
        if (B == A) *nrec_out = 0;
    }
    if (!ok) return;
    const uint32_t n = *ncand_p, nodes = n + 1;
    if (v >= nodes) return;
    uint64_t s = (v == 0) ? A : cands[v - 1];
    uint32_t k = 0, nx = kParEnd;
    uint64_t e = B;
    if (s >= B || (v && s <= A)) {
        J[v] = kParEnd;
        R[v] = 0;
        endpos[v] = B;
        return;
    }
    uint32_t lo = v;
    for (;;) {
        const uint64_t tlo = s + effmin, thi = s + maxsz;
        uint32_t hi = n;
        // the next cut is close: gallop before the binary search (keeps the search inside a few cache lines)
        uint32_t step = 1;
        while (lo + step < n && cands[lo + step] < tlo) { lo += step; step <<= 1; }
        hi = min(n, lo + step + 1);
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (cands[mid] < tlo) lo = mid + 1; else hi = mid;
        }
        const uint64_t c = (lo < n) ? cands[lo] : ~0ull;
        if (c < thi && c <= B) { e = c; nx = (c < B) ? lo + 1 : kParEnd; ++k; break; }
        if (thi >= B) { e = B; nx = kParEnd; ++k; break; }
        if (lo >= n) { k += (uint32_t)((B - s + maxsz - 1) / maxsz); e = B; nx = kParEnd; break; }
        s = thi;
        ++k;
    }
    J[v] = nx;
    R[v] = k;
    endpos[v] = e;
}

__global__ __launch_bounds__(256) void k_par_double(const uint32_t *Jp, const uint32_t *Rp, uint32_t *Jm, uint32_t *Rm,
                                                    const uint32_t *ncand_p, uint32_t node_cap, const pbsgpu_segment *segs) {
    if (!par_active(ncand_p, node_cap) || segs[0].length == 0) return;
    const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= *ncand_p + 1) return;
    const uint32_t a = Jp[v];
    if (a == kParEnd) {
        Jm[v] = kParEnd;
        Rm[v] = Rp[v];
    } else {
        Jm[v] = Jp[a];
        Rm[v] = Rp[v] + Rp[a];
    }
}

__global__ __launch_bounds__(64) void k_par_count(const uint32_t *J, const uint32_t *R, uint32_t levels, uint32_t node_cap,
                                                  const uint32_t *ncand_p, const pbsgpu_segment *segs, uint32_t *nrec_out,
                                                  uint32_t *hops_out) {
    if (threadIdx.x) return;
    if (!par_active(ncand_p, node_cap) || segs[0].length == 0) { *hops_out = 0; return; }
    uint32_t pos = 0, hops = 0, total = 0;
    for (int m = (int)levels - 1; m >= 0; --m) {
        const uint32_t a = J[(size_t)m * node_cap + pos];
        if (a != kParEnd) {
            total += R[(size_t)m * node_cap + pos];
            hops += 1u << m;
            pos = a;
        }
    }
    *hops_out = hops + 1;
    *nrec_out = total + R[pos];
}

__global__ __launch_bounds__(256) void k_par_emit(const uint64_t *cands, const pbsgpu_segment *segs, uint32_t maxsz,
                                                  const uint32_t *J, const uint32_t *R, const uint64_t *endpos, uint32_t levels,
                                                  uint32_t node_cap, const uint32_t *hops_p, pbsgpu_record *recs, uint64_t rec_cap) {
    const uint32_t hops = *hops_p;
    const uint64_t A = segs[0].offset;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t h = blockIdx.x * blockDim.x + threadIdx.x; h < hops; h += stride) {
        uint32_t pos = 0, off = 0;
        for (int m = (int)levels - 1; m >= 0; --m)
            if (h & (1u << m)) {
                off += R[(size_t)m * node_cap + pos];
                pos = J[(size_t)m * node_cap + pos];
            }
        uint64_t s = (pos == 0) ? A : cands[pos - 1];
        const uint32_t k = R[pos];
        const uint64_t e = endpos[pos];
        for (uint32_t t = 0; t < k; ++t) {
            const uint64_t end = (t + 1 < k) ? s + maxsz : e;
            if ((uint64_t)off + t < rec_cap) {
                pbsgpu_record *r = recs + off + t;
                r->end = end - A;
                r->segment = 0;
                r->size = (uint32_t)(end - s);
            }
            s = end;
        }
    }
}

hipError_t launch_resolve_single_par_grid(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                          uint32_t effmin, uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec,
                                          pbsgpu_record *recs, uint64_t rec_cap, void *scratch, uint32_t node_cap, uint32_t levels,
                                          uint32_t *fallback, uint32_t *hops, hipStream_t st) {
    uint32_t *J = static_cast<uint32_t *>(scratch);
    uint32_t *R = J + (size_t)node_cap * levels;
    uint64_t *endpos = reinterpret_cast<uint64_t *>(R + (size_t)node_cap * levels);
    const unsigned nb = (node_cap + 255) / 256;
    hipLaunchKernelGGL(k_par_next, dim3(nb), dim3(256), 0, st, cands, ncand, segs, effmin, maxsz, J, R, endpos, node_cap, fallback,
                       nrec);
    for (uint32_t m = 1; m < levels; ++m)
        hipLaunchKernelGGL(k_par_double, dim3(nb), dim3(256), 0, st, J + (size_t)(m - 1) * node_cap, R + (size_t)(m - 1) * node_cap,
                           J + (size_t)m * node_cap, R + (size_t)m * node_cap, ncand, node_cap, segs);
    hipLaunchKernelGGL(k_par_count, dim3(1), dim3(64), 0, st, (const uint32_t *)J, (const uint32_t *)R, levels, node_cap, ncand, segs,
                       nrec, hops);
    hipLaunchKernelGGL(k_par_emit, dim3(std::min(nb, 4096u)), dim3(256), 0, st, cands, segs, maxsz, (const uint32_t *)J,
                       (const uint32_t *)R, (const uint64_t *)endpos, levels, node_cap, (const uint32_t *)hops, recs, rec_cap);
    // the serial walk, gated: runs only if there were more candidates than nodes (*fallback != 0)
    hipLaunchKernelGGL((k_resolve<true>), dim3(1), dim3(64), 0, st, cands, ncand, segs, 1u, effmin, maxsz, nrec, zero_off, recs,
                       rec_cap, (const uint64_t *)nullptr, (const uint32_t *)nullptr, 0u, (const uint32_t *)fallback,
                       SuggFeed{1, 0, 0, 0}, DenseTiles{}, DenseSeg{});
    return hipGetLastError();
}

size_t resolve_par_scratch_bytes(uint32_t node_cap, uint32_t levels) {
    return (size_t)node_cap * levels * 8 + (size_t)node_cap * 8 + 256;
}

hipError_t launch_resolve_single_par(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs, uint32_t effmin,
                                     uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec, pbsgpu_record *recs,
                                     uint64_t rec_cap, void *scratch, uint32_t node_cap, uint32_t levels, uint32_t *fallback,
                                     hipStream_t st) {
    uint32_t *J = static_cast<uint32_t *>(scratch);
    uint32_t *R = J + (size_t)node_cap * levels;
    uint64_t *endpos = reinterpret_cast<uint64_t *>(R + (size_t)node_cap * levels);
    hipLaunchKernelGGL(k_resolve_par, dim3(1), dim3(1024), 0, st, cands, ncand, segs, effmin, maxsz, nrec, recs, rec_cap, J, R,
                       endpos, node_cap, levels, fallback);
    // the serial walk, gated: runs only if the parallel kernel handed the job back (*fallback != 0)
    hipLaunchKernelGGL((k_resolve<true>), dim3(1), dim3(64), 0, st, cands, ncand, segs, 1u, effmin, maxsz, nrec, zero_off, recs,
                       rec_cap, (const uint64_t *)nullptr, (const uint32_t *)nullptr, 0u, (const uint32_t *)fallback,
                       SuggFeed{1, 0, 0, 0}, DenseTiles{}, DenseSeg{});
    return hipGetLastError();
}

// one segment: its records start at index 0, so a single walk writes them and the count (-> *nrec)
hipError_t launch_resolve_single(const uint64_t *cands, const uint32_t *ncand, const pbsgpu_segment *segs,
                                 uint32_t effmin, uint32_t maxsz, const uint32_t *zero_off, uint32_t *nrec,
                                 pbsgpu_record *recs, uint64_t rec_cap, const Suggested &sg, hipStream_t st,
                                 const DenseTiles *dz, uint32_t lead, uint64_t ntiles) {
    hipLaunchKernelGGL((k_resolve<true>), dim3(1), dim3(64), 0, st, cands, ncand, segs, 1u, effmin, maxsz, nrec,
                       zero_off, recs, rec_cap, sg.offsets, sg.index, sg.cmin, (const uint32_t *)nullptr,
                       SuggFeed{sg.feed, sg.origin, sg.absolute, sg.open_end}, dz ? *dz : DenseTiles{},
                       flat_dense_seg(dz, lead, ntiles));
    return hipGetLastError();
}

// =====================================================================================
// (3) SHA-256 — one lane per byte range, lanes pull ranges from a queue
// =====================================================================================
__device__ __forceinline__ uint32_t rotr(uint32_t x, int n) { return __builtin_rotateright32(x, n); }

// gfx950 has v_bitop3_b32 (any 3-input bitwise op in one instruction): XOR3 = 0x96, MAJ = 0xE8.
__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
}
__device__ __forceinline__ uint32_t ch3(uint32_t e, uint32_t f, uint32_t g) { return g ^ (e & (f ^ g)); }  // v_bfi

__device__ constexpr uint32_t kSha256K[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

// message schedule step: W[i & 15] <- sigma1(W[i-2]) + W[i-7] + sigma0(W[i-15]) + W[i-16]
__device__ __forceinline__ uint32_t sha256_sched(uint32_t (&W)[16], int i) {
    const uint32_t w15 = W[(i + 1) & 15], w2 = W[(i + 14) & 15];
    const uint32_t s0 = xor3(rotr(w15, 7), rotr(w15, 18), w15 >> 3);
    const uint32_t s1 = xor3(rotr(w2, 17), rotr(w2, 19), w2 >> 10);
    const uint32_t w = W[i & 15] + s0 + W[(i + 9) & 15] + s1;
    W[i & 15] = w;
    return w;
}

// one round given wk = W[t] + K[t]; 14 VALU instructions (3+1 Sigma1, bfi, 2 adds, 3+1 Sigma0, maj, 2 adds)
#define SHA256_ROUND(a, b, c, d, e, f, g, h, wk)                                   \
    do {                                                                           \
        const uint32_t t1_ = (h) + xor3(rotr(e, 6), rotr(e, 11), rotr(e, 25)) + ch3(e, f, g) + (wk); \
        const uint32_t t2_ = xor3(rotr(a, 2), rotr(a, 13), rotr(a, 22)) + maj3(a, b, c); \
        (d) += t1_;                                                                \
        (h) = t1_ + t2_;                                                           \
    } while (0)

__device__ __forceinline__ void sha256_compress(uint32_t (&H)[8], uint32_t (&W)[16]) {
    uint32_t a = H[0], b = H[1], c = H[2], d = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
#pragma unroll
    for (int i = 0; i < 64; i += 8) {
        uint32_t wk[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) wk[j] = ((i + j < 16) ? W[i + j] : sha256_sched(W, i + j)) + kSha256K[i + j];
        SHA256_ROUND(a, b, c, d, e, f, g, h, wk[0]);
        SHA256_ROUND(h, a, b, c, d, e, f, g, wk[1]);
        SHA256_ROUND(g, h, a, b, c, d, e, f, wk[2]);
        SHA256_ROUND(f, g, h, a, b, c, d, e, wk[3]);
        SHA256_ROUND(e, f, g, h, a, b, c, d, wk[4]);
        SHA256_ROUND(d, e, f, g, h, a, b, c, wk[5]);
        SHA256_ROUND(c, d, e, f, g, h, a, b, wk[6]);
        SHA256_ROUND(b, c, d, e, f, g, h, a, wk[7]);
    }
    H[0] += a; H[1] += b; H[2] += c; H[3] += d; H[4] += e; H[5] += f; H[6] += g; H[7] += h;
}

__device__ __forceinline__ void sha256_iv(uint32_t (&H)[8]) {
    H[0] = 0x6a09e667; H[1] = 0xbb67ae85; H[2] = 0x3c6ef372; H[3] = 0xa54ff53a;
    H[4] = 0x510e527f; H[5] = 0x9b05688c; H[6] = 0x1f83d9ab; H[7] = 0x5be0cd19;
}

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

// Chunk data is read through GLOBAL-address-space pointers, never generic ones. A generic pointer compiles to flat_load,
// which counts in lgkmcnt as well as vmcnt; the producer's `s_waitcnt lgkmcnt(0)` in front of every s_barrier (its LDS
// writes must have landed) then also waits for the block it requested a moment ago, i.e. the FIFO's prefetch distance
// collapses to less than one step and the step time follows the slowest of the wave's 320 outstanding requests
// (rounds 3-5 shipped that; the address space is not inferred through the lambdas and the page select). global_load
// counts in vmcnt only.
#define PBSK_GLOBAL __attribute__((address_space(1)))
typedef const PBSK_GLOBAL u32x4_a4 *gvec4_ptr;
typedef const PBSK_GLOBAL uint32_t *gword_ptr;

// What a producer lane WITHOUT a data block requests in a step (and ignores). The request itself is unconditional, FIVE
// loads in every step of every lane: only then does the number of requests issued after a slot's own five not depend on
// the path taken, and the compiler can await a slot with `s_waitcnt vmcnt(5)` — the other slot's requests stay in flight
// — instead of vmcnt(0). (A branch around the loads, even a wave-uniform one, brings the vmcnt(0) back.)
__device__ uint32_t kIdleBlock[32];

// Request the 64 bytes at p (+ the 4 behind them if p is not 4-byte aligned) as raw little-endian dwords; `sel` = the
// v_perm selector that byte-swaps and funnels them into big-endian message words.
__device__ __forceinline__ void sha256_request_block(const uint8_t *p, uint32_t (&R)[17], uint32_t &sel) {
    const uint32_t o = (uint32_t)((uintptr_t)p & 3u);
    const gvec4_ptr q = (gvec4_ptr)(p - o);
    const u32x4_a4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
    R[0] = v0.x; R[1] = v0.y; R[2] = v0.z; R[3] = v0.w;
    R[4] = v1.x; R[5] = v1.y; R[6] = v1.z; R[7] = v1.w;
    R[8] = v2.x; R[9] = v2.y; R[10] = v2.z; R[11] = v2.w;
    R[12] = v3.x; R[13] = v3.y; R[14] = v3.z; R[15] = v3.w;
    R[16] = ((gword_ptr)(p - o))[o ? 16 : 15];  // aligned: nothing behind the block is touched (the word is not selected)
    sel = ((o) << 24) | ((o + 1) << 16) | ((o + 2) << 8) | (o + 3);
}

// Raw little-endian words of a block that is NOT 64 full data bytes (the last data bytes + 0x80 marker, zero fill, and
// the big-endian bit length in the final block): the funnel-shifted data words come from aligned dword loads whose
// addresses are clamped to the range's last valid dword (nothing beyond the chunk is touched), then everything behind the
// data is masked off and the marker / length are or-ed in. ~130 instructions with 17 independent loads in flight — the
// byte-at-a-time assembly this replaces cost dozens of DEPENDENT byte loads per tail, which at small chunk sizes (a tail
// every ~1000 blocks per lane, i.e. one per ~16 wave iterations) is what held the hash kernels back.
__device__ __forceinline__ void sha256_tail_words(const uint8_t *base, uint64_t len, uint64_t off, bool last, uint32_t (&R)[17]) {
    const int64_t rem = (int64_t)len - (int64_t)off;                // data bytes from `off` on (<= 0: padding-only block)
    const uint32_t valid = rem <= 0 ? 0u : (rem >= 64 ? 64u : (uint32_t)rem);
    uint32_t q[17];
#pragma unroll
    for (int j = 0; j < 17; ++j) q[j] = 0;
    uint32_t o = 0;
    if (valid) {
        const uint8_t *p = base + off;
        o = (uint32_t)((uintptr_t)p & 3u);
        const gword_ptr a = (gword_ptr)(p - o);
        const uint32_t jmax = (o + valid - 1u) >> 2;               // last dword that holds a valid byte
#pragma unroll
        for (int j = 0; j < 17; ++j) q[j] = a[min((uint32_t)j, jmax)];
    }
    const uint32_t sh = o * 8u;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        uint32_t w = sh ? ((q[j] >> sh) | (q[j + 1] << (32u - sh))) : q[j];  // bytes off+4j .. off+4j+3, little-endian
        const int32_t k = (int32_t)valid - 4 * j;                  // valid bytes in this word
        const uint32_t mask = k >= 4 ? 0xffffffffu : (k <= 0 ? 0u : ((1u << (8 * k)) - 1u));
        w &= mask;
        if (rem >= 0 && rem < 64 && k >= 0 && k < 4) w |= 0x80u << (8 * k);  // the marker byte sits at position len
        R[j] = w;
    }
    if (last) {  // big-endian bit length in bytes 56..63 (R is little-endian raw)
        const uint64_t bits = len * 8;
        R[14] = __builtin_bswap32((uint32_t)(bits >> 32));
        R[15] = __builtin_bswap32((uint32_t)bits);
    }
    R[16] = 0;
}

// Work source for the SHA kernel: item i -> (byte pointer, length, digest destination)
struct RecordSource {
    static constexpr bool kRing = false;
    pbsgpu_record *recs;
    // queue position -> {chunk address lo, hi, size, record index}, longest chunks first (k_order). ONE 16-byte load
    // per chunk a lane takes from the queue: the chain order[i] -> record -> segment offset was three dependent round
    // trips (~1-2 us each under load) at every chunk boundary of every lane, and all pairs of a workgroup wait at the
    // block barrier while one producer sits in them.
    const uint4 *qdesc;
    __device__ __forceinline__ void get(uint32_t i, const uint8_t *&ptr, uint64_t &len, uint8_t *&dst) const {
        const uint4 d = qdesc[i];
        ptr = reinterpret_cast<const uint8_t *>(((uint64_t)d.y << 32) | d.x);
        len = d.z;
        dst = recs[d.w].digest;
    }
};
struct SegmentSource {
    static constexpr bool kRing = false;
    const uint8_t *data;
    const pbsgpu_segment *segs;
    uint8_t *digests;
    __device__ __forceinline__ void get(uint32_t i, const uint8_t *&ptr, uint64_t &len, uint8_t *&dst) const {
        ptr = data + segs[i].offset;
        len = segs[i].length;
        dst = digests + (uint64_t)i * 32;
    }
};

// The page ring's SHA-256 SERVICE (ring.cpp): one persistent launch whose lanes pull chunk descriptors from a
// device-resident FIFO that the cut rounds of ALL streams append to (k_ring_order / k_ring_publish), so a chunk starts
// hashing the moment it has been cut, whichever round, stream or page it came from, and every lane that finishes a chunk
// takes the next one — no batch-granular makespan, no batch-granular buffer release.
//   * a lane CLAIMS queue positions with one wave-level atomicAdd on `head` and then waits for `tail` to pass its
//     position (positions are unique and every position is eventually published, or `stop` is raised);
//   * a chunk may cross from one physical page into another (pages of a stream are not adjacent): the descriptor
//     carries both piece addresses; a block that starts in the first piece reads on into the page's 128-byte tail pad,
//     which mirrors the first bytes of the stream's next page;
//   * when a lane has LOADED the last block of a chunk it drops the chunk's reference on its page(s); the lane that
//     brings a page to zero reports it to the host through a FIFO in mapped pinned memory — the page can take new
//     bytes while the consumer wave is still compressing the chunk's last blocks;
//   * the consumer writes the digest straight into the host-visible record cell and raises the cell's flag.
// (RingCtl / RingSource: kernels.h)

// the services' regime probe (RingSource::probe): called once per block step by the service's probe wave, wave-uniform
struct RingProbe {
    uint32_t n = 0, idle = 0, act = 0;
    unsigned long long c0 = 0, w0 = 0;
};
// (`active`: lanes of the probe wave that carried a block in this step -> probe[8] for the pair service: how full its lanes are,
// pbsgpu_ring_debug)
__device__ __forceinline__ void ring_probe_step(const RingSource &src, RingProbe &pb, const bool busy, const int slot, const int lane,
                                                const uint32_t active = 0) {
    pb.idle |= busy ? 0u : 1u;
    pb.act += active;
    if (++pb.n < kRingProbeSteps) return;
    const unsigned long long c = clock64(), w = wall_clock64();
    if (!pb.idle && pb.w0 != 0ull && lane == 0 && src.probe) {
        atomicAdd(src.probe + slot, (unsigned long long)kRingProbeSteps);
        atomicAdd(src.probe + slot + 1, c - pb.c0);
        atomicAdd(src.probe + slot + 2, w - pb.w0);
        if (slot == 0) atomicAdd(src.probe + 8, (unsigned long long)pb.act);
    }
    pb.n = 0;
    pb.act = 0;
    pb.idle = 0;
    pb.c0 = c;
    pb.w0 = w;
}

// Each lane streams one byte range through SHA-256. Per loop trip every busy lane consumes
// one 64-byte block: the raw dwords of the NEXT block are requested before the current
// block is compressed, so the HBM/L2 latency of a lane's private stream hides behind the
// ~1700 VALU ops of the compression even with a single wave on the SIMD.
template <typename Source>
__global__ __launch_bounds__(64) void k_sha256(Source src, const uint32_t *nitems_p, uint32_t nitems_imm,
                                               uint32_t *queue, const uint32_t *wg_limit, uint32_t whole_chip) {
    // k_order's budget counts 128-lane workgroups of the pair kernel = two of these waves; waves beyond it leave
    // their SIMD slot to the other batches in flight (the queue is dynamic, the remaining waves drain it).
    // whole_chip (dense launches): every wave works unless k_order said "skip this pass" (0)
    if (wg_limit) {
        const uint32_t lim = *wg_limit;
        if (whole_chip ? lim == 0 : blockIdx.x >= lim * 2u) return;
    }
    const int lane = threadIdx.x & 63;
    const uint32_t nitems = nitems_p ? *nitems_p : nitems_imm;

    // "next" block descriptor (what R holds / will hold)
    const uint8_t *base = nullptr;  // start of the lane's byte range
    uint64_t len = 0;               // its length
    uint64_t blk = 0, nblk = 0;     // index of the block in R, total blocks incl. padding
    uint8_t *dst = nullptr;
    bool have = false;              // R holds a valid block
    bool exhausted = false;         // queue is empty for this lane

    uint32_t R[17];                 // raw little-endian dwords of the next block (+1 for misalignment)
    uint32_t sel = 0x00010203u;     // v_perm selector: byte swap + misalignment shift
#pragma unroll
    for (int j = 0; j < 17; ++j) R[j] = 0;

    uint32_t H[8];
    sha256_iv(H);

    // loads block `blk` of (base,len) into R (raw) and sets sel
    auto load_block = [&]() {
        const uint64_t off = blk * 64;
        if (off + 64 <= len) {  // pure data block: aligned dword loads + per-lane funnel selector
            const uint8_t *p = base + off;
            const uint32_t o = (uint32_t)((uintptr_t)p & 3u);
            const gvec4_ptr q = (gvec4_ptr)(p - o);
            const u32x4_a4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
            R[0] = v0.x; R[1] = v0.y; R[2] = v0.z; R[3] = v0.w;
            R[4] = v1.x; R[5] = v1.y; R[6] = v1.z; R[7] = v1.w;
            R[8] = v2.x; R[9] = v2.y; R[10] = v2.z; R[11] = v2.w;
            R[12] = v3.x; R[13] = v3.y; R[14] = v3.z; R[15] = v3.w;
            R[16] = o ? ((gword_ptr)(p - o))[16] : 0u;
            sel = ((o) << 24) | ((o + 1) << 16) | ((o + 2) << 8) | (o + 3);
        } else {  // tail / padding block (at most two per range)
            sha256_tail_words(base, len, off, blk + 1 == nblk, R);
            sel = 0x00010203u;
        }
    };

    // acquire a new range for lanes that need one
    auto acquire = [&](bool need) {
        const unsigned long long m = __ballot(need);
        if (m == 0) return;
        uint32_t first = 0;
        const int leader = __ffsll((long long)m) - 1;
        if (lane == leader) first = atomicAdd(queue, (uint32_t)__popcll(m));
        first = __shfl(first, leader, 64);
        if (need) {
            const uint32_t i = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (i < nitems) {
                src.get(i, base, len, dst);
                blk = 0;
                nblk = (len + 8) / 64 + 1;
                have = true;
            } else {
                exhausted = true;
                have = false;
            }
        }
    };

    acquire(true);
    if (have) load_block();

    while (__any(have)) {
        // current block = R; convert to big-endian message words
        uint32_t W[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) W[j] = __builtin_amdgcn_perm(R[j + 1], R[j], sel);
        const bool cur = have;
        const bool cur_last = have && (blk + 1 == nblk);
        uint8_t *cur_dst = dst;

        // advance to the next block / next range and request its bytes
        if (have) {
            if (!cur_last) ++blk; else have = false;
        }
        acquire(!have && !exhausted);
        if (have) load_block();

        if (cur) sha256_compress(H, W);

        if (cur_last) {
            uint32_t *o = reinterpret_cast<uint32_t *>(cur_dst);  // digest is 4-byte aligned in both sources
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(H[j]);
            sha256_iv(H);
        }
    }
}

// -------------------------------------------------------------------------------------
// SHA-256, LANES service of the page ring (RingSource::sdesc): the single-wave form above on the ring's protocol. One lane per
// chunk, schedule and rounds in the same wave, no LDS hand-over and no barrier: four independent waves per CU (the launch takes
// the CU's whole LDS like the other services). 81 chain-blocks per us and CU against the pair form's 74 — and 3.2 us per block for
// every chain, which is why only SHORT chunks come here and only while this service has room (k_ring_control). The queue is a
// compare-and-swap queue like the long one: a lane never claims a position it then has to wait at.
__global__ __launch_bounds__(512) void k_sha256_lanes(RingSource src) {
    const int lane = threadIdx.x & 63;
    const uint8_t *base = nullptr, *base2 = nullptr;
    uint64_t len = 0, blk = 0, nblk = 0;
    uint32_t len1 = 0, pages = 0xffffffffu, poll_ctr = 0;
    uint8_t *dst = nullptr;
    bool have = false, exhausted = false;
    uint32_t R[17];
    uint32_t sel = 0x00010203u;
#pragma unroll
    for (int j = 0; j < 17; ++j) R[j] = 0;
    uint32_t H[8];
    sha256_iv(H);

    auto load_block = [&]() {  // request block `blk` of the lane's chunk (k_sha256_pair's prep)
        const uint64_t off = blk * 64;
        const uint8_t *bb = (off < len1) ? base : base2;  // which physical page holds this block
        if (off + 64 <= len) {
            sha256_request_block(bb + off, R, sel);
        } else {
            sha256_tail_words(bb, len, off, blk + 1 == nblk, R);
            sel = 0x00010203u;
        }
    };
    auto acquire = [&](bool need) {
        if (__ballot(need) == 0) return;
        if (__ballot(have) != 0 && ((++poll_ctr) & src.poll_mask) != 0u) return;
        const unsigned long long sq = __hip_atomic_load(&src.ctl->sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t st = (uint32_t)sq, sh = (uint32_t)(sq >> 32);
        const int32_t avail = (int32_t)(st - sh);
        const unsigned long long mn = __ballot(need);
        if (avail > 0) {
            const uint32_t cnt = min((uint32_t)__popcll(mn), (uint32_t)avail);
            const int leader = __ffsll((long long)mn) - 1;
            uint32_t got0 = 0xffffffffu;
            if (lane == leader && atomicCAS(&src.ctl->shead, sh, sh + cnt) == sh) got0 = sh;
            got0 = __shfl(got0, leader, 64);
            if (got0 != 0xffffffffu) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                const uint32_t rank = (uint32_t)__popcll(mn & ((1ull << lane) - 1ull));
                const bool got = need && rank < cnt;
                uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(0, 0, 0, 0);
                if (got) {
                    e0 = src.sdesc[2u * ((got0 + rank) & src.smask)];
                    e1 = src.sdesc[2u * ((got0 + rank) & src.smask) + 1u];
                }
                base = got ? reinterpret_cast<const uint8_t *>(((uint64_t)e0.y << 32) | e0.x) : base;
                base2 = got ? reinterpret_cast<const uint8_t *>(((uint64_t)e1.y << 32) | e1.x) : base2;
                len = got ? (uint64_t)e0.z : len;
                len1 = got ? e0.w : len1;
                dst = got ? src.cells + (uint64_t)e1.z * 64u + 8u : dst;
                pages = got ? e1.w : pages;
                blk = got ? 0ull : blk;
                nblk = got ? ((uint64_t)e0.z + 8u) / 64u + 1u : nblk;
                have = have | got;
            }
        } else {
            // nothing published: has the service been told to stop? (stop is raised behind the last publish; the queue is looked
            // at again behind the fence, so no lane leaves while short chunks are unclaimed)
            const unsigned long long ts = __hip_atomic_load(&src.ctl->tail_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(ts >> 32) != 0u) {
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                const unsigned long long sq2 = __hip_atomic_load(&src.ctl->sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((int32_t)((uint32_t)sq2 - (uint32_t)(sq2 >> 32)) <= 0) exhausted = exhausted | need;
            }
        }
    };

    // regime probe of this service (pbsgpu_ring_debug): workgroup 0's first wave, steps and 100 MHz ticks of the intervals in which
    // it carried a block in every step -> probe[6], probe[7]
    const bool probe_on = src.probe != nullptr && blockIdx.x == 0 && threadIdx.x < 64;
    uint32_t pn = 0, pidle = 0, pact = 0;
    unsigned long long pw0 = 0;
    acquire(true);
    if (have) load_block();
    for (;;) {
        if (probe_on) {
            const unsigned long long hm = __ballot(have);
            pidle |= hm ? 0u : 1u;
            pact += (uint32_t)__popcll(hm);
            if (++pn >= kRingProbeSteps) {
                const unsigned long long w = wall_clock64();
                if (!pidle && pw0 != 0ull && lane == 0) {
                    atomicAdd(src.probe + 6, (unsigned long long)kRingProbeSteps);
                    atomicAdd(src.probe + 7, w - pw0);
                    atomicAdd(src.probe + 9, (unsigned long long)pact);
                }
                pn = 0; pidle = 0; pact = 0; pw0 = w;
            }
        }
        if (!__any(have)) {  // the wave holds nothing: leave once every lane has seen `stop` with the queue empty, else nap and look again
            if (__ballot(!exhausted) == 0ull) break;
            __builtin_amdgcn_s_sleep(48);
            poll_ctr = 0;
            acquire(!exhausted);
            if (have) load_block();
            continue;
        }
        uint32_t W[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) W[j] = __builtin_amdgcn_perm(R[j + 1], R[j], sel);
        const bool cur = have;
        const bool cur_last = have && (blk + 1 == nblk);
        uint8_t *cur_dst = dst;
        if (have) {
            if (!cur_last) ++blk; else have = false;
        }
        if (cur_last) {  // the chunk's last block is in registers: drop its page references (k_sha256_pair)
            const uint32_t pg = pages;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t pi = h ? (pg >> 16) : (pg & 0xffffu);
                if (pi != 0xffffu) {
                    const uint32_t old = __hip_atomic_fetch_sub(&src.pending[pi], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    if (old == 1u) {
                        const uint32_t fs = atomicAdd(&src.ctl->free_count, 1u);
                        __hip_atomic_store(&src.free_fifo[fs & src.free_mask], ((unsigned long long)(fs + 1u) << 32) | pi,
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                }
            }
        }
        acquire(!have && !exhausted);
        if (have) load_block();
        if (cur) sha256_compress(H, W);
        if (cur_last) {  // record cell in mapped pinned memory: digest first, then its flag
            PBSK_GLOBAL uint32_t *o = (PBSK_GLOBAL uint32_t *)cur_dst;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(H[j]);
            __threadfence_system();
            __hip_atomic_store(o + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            sha256_iv(H);
        }
    }
}

// -------------------------------------------------------------------------------------
// SHA-256, wave-pair form. The ingest workload is parallelism-starved on this chip: at a
// 4 MiB average chunk even a full 288 GB of HBM holds ~70 k chunks, while 256 CUs x 4 SIMDs
// x 2 waves offer 131 k lanes, and a lone wave only issues one instruction every ~4 cycles.
// A batch therefore finishes when its LONGEST chunk does (max = 16 MiB = 262 144 dependent
// compressions), so what matters is the number of instructions on that serial chain. A
// workgroup is two waves on two SIMDs: the PRODUCER wave owns the byte streams (queue,
// loads, tail/padding, byte swap, message schedule, + K) and hands W[t]+K[t] through LDS;
// the CONSUMER wave executes nothing but the 64 rounds (14 VALU each) and the digest store.
// One s_barrier per block; LDS double-buffered (2 x 16 KiB).
// DENSE form (second instantiation, chosen per launch by the host: sha256_dense_pays): when a job holds more work than
// the two pairs can finish within its longest chain, the chain no longer bounds the job — issue slots do, and a pair
// leaves them unused: the producers' SIMDs idle part of every step. The workgroup is then EIGHT waves (4 pairs):
// waves 4,5 = producers of pairs 2,3 (they land on the SIMDs of consumers 0,1), waves 6,7 = consumers of pairs 2,3
// (on the SIMDs of producers 0,1; placement verified with HW_ID, profiles/r02_probe_simd_placement.log), so every SIMD
// hosts one consumer and one producer. The gain is bounded by the instruction counts: per block the consumer issues
// ~935 instructions and the producer ~860 (744 VALU: 48 x 11 schedule + K adds + perms + addressing), so four pairs
// per ~1800 issue slots against two pairs per ~935 is +7 % at best; measured +5 % (SHA alone 65.3 -> 62.0 ms for
// 64 GiB at 64 KiB average). Each chain is ~1.9x slower in this form, so it is used only when the chain does not bound.
template <typename Source, bool DENSE>
__global__ __launch_bounds__(DENSE ? 512 : 256) void k_sha256_pair(Source src, const uint32_t *nitems_p, uint32_t nitems_imm,
                                                                   uint32_t *queue, const uint32_t *wg_limit) {
    // The HOST picks the form per launch (sha256_dense_pays: from the batch's byte count and the chunker's maximum —
    // k_order could decide more exactly on the device, but then both forms have to be enqueued and the idle one still
    // waits for CUs with room for its LDS before it can leave: measured 2.7 ms avg / 14 ms max per batch on the default
    // workload, 46 ms at 64 KiB average). Two instantiations rather than a run-time switch: launch bounds and LDS
    // differ, and the sparse code stays exactly what was tuned (one 512-thread kernel with a switch ran the sparse chain
    // at 463-465 ms per 64 GiB against 433-458; within box-to-box spread). Sparse: workgroups beyond
    // k_order's budget leave at once so their CUs can host another batch's kernel; dense: the budget is the whole
    // chip, only k_order's "skip this pass" (0) is honoured.
    if (wg_limit) {
        const uint32_t lim = *wg_limit;
        if (DENSE ? lim == 0 : blockIdx.x >= lim) return;
    }
    // One workgroup per CU (sparse: 69 KB static LDS + the launch's padding; dense: 137 KB).
    // waves 0,1 = consumers of pair 0,1; waves 2,3 = their producers;
    // dense only: waves 4,5 = producers of pair 2,3; waves 6,7 = their consumers.
    constexpr int NP = DENSE ? 4 : 2;
    __shared__ uint4 wkbuf_[NP][2][16][64];  // [pair][buffer][4 rounds][lane] -> 16 B per lane, contiguous rows
    __shared__ uint32_t ctrl_[NP][2][64];    // bit0 block valid, bit1 last block of its range
    __shared__ uint8_t *dstp_[NP][2][64];    // digest destination (valid when bit1)
    __shared__ uint32_t alive[NP][2];        // [pair][buffer]: producer still had blocks
    __shared__ uint32_t curw[NP][2];         // [pair][buffer]: the pair had a block in this step (ring service: who may nap)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pr = (wave & 1) | (wave >= 4 ? 2 : 0);
    const bool producer = (wave >= 2 && wave < 6);
    auto &wkbuf = wkbuf_[pr];
    auto &ctrl = ctrl_[pr];
    auto &dstp = dstp_[pr];
    auto any_alive = [&](const int pb) -> uint32_t {
        uint32_t a = alive[0][pb] | alive[1][pb];
        if constexpr (DENSE) a |= alive[2][pb] | alive[3][pb];
        return a;
    };

    if (producer) {
        const uint32_t nitems = nitems_p ? *nitems_p : nitems_imm;
        // byte-stream state of this lane
        const uint8_t *base = nullptr;
        uint64_t len = 0, blk = 0, nblk = 0;  // blk = next block to fetch
        uint8_t *dst = nullptr;
        bool have = false, exhausted = false;
        // ring service only: second piece of a chunk that crosses into another physical page (virtual base: the byte at
        // chunk offset `off >= len1` lives at base2 + off), the chunk's page references, the lane's claimed queue position
        [[maybe_unused]] const uint8_t *base2 = nullptr;
        [[maybe_unused]] uint32_t len1 = 0, pages = 0xffffffffu, claim = 0;
        [[maybe_unused]] uint32_t claimed = 0;  // (a word, not a bool: two bool flags set in sibling branches get their stores
                                                // merged through a selected pointer by the optimiser, which puts both in scratch)
        [[maybe_unused]] unsigned long long idle_since = 0;
        [[maybe_unused]] uint32_t hb_seen = 0;
        // FIFO of raw blocks in flight: a block is requested D iterations before it is expanded, so
        // HBM/TLB latency of the lane-private streams stays off the serial chain
        // (round 6: D = 4 in the ring's service — is it memory latency that makes the producer the slower half under load, 1.92-1.95
        // against 1.81-1.82 us per step? No: 604-612 GiB/s against 618-619 on the same box, profiles/r06_ab_restored_services_and_fifo_depth.log)
        constexpr int D = 2;  // even (buffer parity is derived from the slot index)
        uint32_t R[D][17];
        uint32_t selv[D], cflag[D];
        uint8_t *dstv[D];
        [[maybe_unused]] uint32_t pagesv[D];
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int j = 0; j < 17; ++j) R[s][j] = 0;
            selv[s] = 0x00010203u;
            cflag[s] = 0;
            dstv[s] = nullptr;
            pagesv[s] = 0xffffffffu;
        }

        // Dense form: a wave RESERVES kReserve extra queue positions with every atomic and serves its lanes from that
        // reserve, so most chunk boundaries cost no atomic round trip at all (items >> lanes there; a few positions held
        // back by a wave at the very end are handed to its own lanes). The sparse form must not hoard — it has more
        // lanes than chunks, and the longest-first order is what keeps its makespan at the longest chain.
        constexpr uint32_t kReserve = 16;
        uint32_t res_next = 0, res_end = 0;  // the wave's reserved positions [res_next, res_end): same value in every lane
        [[maybe_unused]] uint32_t poll_ctr = 0;
        auto acquire = [&](bool need) {
            if constexpr (Source::kRing) {
                // Lanes without work look at the queue — a device-scope load the wave then WAITS for (0.3-0.5 us: about the
                // slack the producer has against its consumer). With every step of a wave that still carries chunks on its
                // other lanes this made the PRODUCER the slower half of the pair whenever one lane was idle: the chain of a
                // max-size chunk ran at ~1.95 us per block instead of 1.7 in a draining ring (drain floor 0.515 s, one file
                // alone 0.58 s) and in every wave with a starved lane during the feed phase. A wave with work therefore
                // polls on every 8th step only (src.poll_mask, PBSGPU_RING_POLL_EVERY; a free lane waits <= 14 us for its next
                // chunk: 0.02 % of a 4 MiB chunk's chain); a wave with no work at all polls every step, as before.
                if (__ballot(need) == 0) return;
                if (__ballot(have) != 0 && ((++poll_ctr) & src.poll_mask) != 0u) return;
                // (0) LONG chunks first (see RingSource::ldesc): one relaxed load of {ltail, lhead}; the wave's leader moves
                // lhead forward by compare-and-swap only over published positions, so no lane ever waits on this queue
                // Who may take a long chunk: a lane that holds NO claim on the main queue — one that has just finished a
                // chunk, or one of the few lanes (1 in 16) that never claim there. A lane that waits at a claimed, not yet
                // published position must not: the chunk published at its position later would then wait for the whole
                // long chain (measured: +0.26-0.40 s on a single file when every idle lane could take long chunks).
                const bool long_only = src.long_bytes != 0u && src.xp == 0u && (lane & 15) == 0;
                const bool elig = need && claimed == 0u;
                if (src.long_bytes && __ballot(elig)) {
                    const unsigned long long lq = __hip_atomic_load(&src.ctl->lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t lt = (uint32_t)lq, lh = (uint32_t)(lq >> 32);
                    int32_t avail = (int32_t)(lt - lh) - (int32_t)src.long_spill;  // (spill: what the express service's lanes are left)
                    // with an express service: only while ALL its lane pairs are busy — a long chunk that waited for one would
                    // finish later than on a pair lane that is free now (configs[2]: half the bytes are 16 MiB chunks)
                    // (safety valve: more long chunks waiting than the express service has pairs — e.g. its workgroups have not
                    // found their CUs yet — are everybody's business)
                    if (avail > 0 && src.xp != 0u && (uint32_t)avail <= src.xp_pairs &&
                        __hip_atomic_load(&src.ctl->xp_busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < src.xp_pairs)
                        avail = 0;
                    if (avail > 0) {
                        const unsigned long long mn = __ballot(elig);
                        const uint32_t cnt = min((uint32_t)__popcll(mn), (uint32_t)avail);
                        const int leader = __ffsll((long long)mn) - 1;
                        uint32_t got0 = 0xffffffffu;
                        if (lane == leader && atomicCAS(&src.ctl->lhead, lh, lh + cnt) == lh) got0 = lh;
                        got0 = __shfl(got0, leader, 64);
                        if (got0 != 0xffffffffu) {
                            __atomic_thread_fence(__ATOMIC_ACQUIRE);
                            const uint32_t rank = (uint32_t)__popcll(mn & ((1ull << lane) - 1ull));
                            const bool tk = elig && rank < cnt;
                            uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(0, 0, 0, 0);
                            if (tk) {
                                e0 = src.ldesc[2u * ((got0 + rank) & src.lmask)];
                                e1 = src.ldesc[2u * ((got0 + rank) & src.lmask) + 1u];
                            }
                            base = tk ? reinterpret_cast<const uint8_t *>(((uint64_t)e0.y << 32) | e0.x) : base;
                            base2 = tk ? reinterpret_cast<const uint8_t *>(((uint64_t)e1.y << 32) | e1.x) : base2;
                            len = tk ? (uint64_t)e0.z : len;
                            len1 = tk ? e0.w : len1;
                            dst = tk ? src.cells + (uint64_t)e1.z * 64u + 8u : dst;
                            pages = tk ? e1.w : pages;
                            blk = tk ? 0ull : blk;
                            nblk = tk ? ((uint64_t)e0.z + 8u) / 64u + 1u : nblk;
                            have = have | tk;
                            need = need && !tk;
                        }
                    }
                }
                // (1) lanes without a position claim one: ONE atomic per wave
                const bool want = need && !claimed && !long_only;
                const unsigned long long mw = __ballot(want);
                if (mw) {
                    uint32_t first = 0;
                    const int leader = __ffsll((long long)mw) - 1;
                    if (lane == leader) {
                        const uint32_t cnt = (uint32_t)__popcll(mw);
                        first = atomicAdd(&src.ctl->head, cnt);
                        // claim progress for the host's backlog gate (ring.cpp): one posted write to mapped pinned
                        // memory whenever the head crosses a multiple of 256
                        if (((first + cnt) ^ first) >> 8)
                            __hip_atomic_store(const_cast<uint32_t *>(src.heartbeat) + 32, first + cnt, __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    first = __shfl(first, leader, 64);
                    if (want) {
                        claim = first + (uint32_t)__popcll(mw & ((1ull << lane) - 1ull));
                        claimed = 1u;
                    }
                }
                // (2) has the queue reached the lane's position? A relaxed device-scope load of {tail, stop} (no cache
                // invalidate per poll); only a lane that really takes a descriptor pays the acquire fence behind which
                // the descriptor and the chunk's bytes (written by other kernels while this one runs) are read.
                if (__ballot(need) == 0) return;
                const unsigned long long ts = __hip_atomic_load(&src.ctl->tail_stop,
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t tail = (uint32_t)ts, stop = (uint32_t)(ts >> 32);
                const bool ready = need && claimed != 0u && (int32_t)(tail - claim) > 0;
                if (__ballot(ready)) __atomic_thread_fence(__ATOMIC_ACQUIRE);
                uint4 d0 = make_uint4(0, 0, 0, 0), d1 = make_uint4(0, 0, 0, 0);
                if (ready) {
                    d0 = src.desc[2u * (claim & src.qmask)];
                    d1 = src.desc[2u * (claim & src.qmask) + 1u];
                }
                // size 0 = a void position (the open chunk of a round): the lane takes another one next time.
                // (selects and or-updates, no conditional stores: two flags set in sibling branches get their stores merged
                // through a selected pointer by the optimiser, which moves both flags into scratch memory)
                const bool got = ready && d0.z != 0u;
                claimed = ready ? 0u : claimed;
                base = got ? reinterpret_cast<const uint8_t *>(((uint64_t)d0.y << 32) | d0.x) : base;
                base2 = got ? reinterpret_cast<const uint8_t *>(((uint64_t)d1.y << 32) | d1.x) : base2;
                len = got ? (uint64_t)d0.z : len;
                len1 = got ? d0.w : len1;
                dst = got ? src.cells + (uint64_t)d1.z * 64u + 8u : dst;
                pages = got ? d1.w : pages;
                blk = got ? 0ull : blk;
                nblk = got ? ((uint64_t)d0.z + 8u) / 64u + 1u : nblk;
                have = have | got;
                // stop: nothing will ever be published at this position. (stop is raised behind the last publish of BOTH queues;
                // the long queue is looked at again after the fence, so a lane never leaves while long chunks are unclaimed)
                bool leave = need && !ready && stop != 0u;
                if (src.long_bytes && src.xp == 0u && __ballot(leave)) {
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    const unsigned long long lq2 = __hip_atomic_load(&src.ctl->lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((int32_t)((uint32_t)lq2 - (uint32_t)(lq2 >> 32)) > 0) leave = false;
                }
                exhausted = exhausted | leave;
                return;
            }
            const unsigned long long m = __ballot(need);
            if (m == 0) return;
            uint32_t first = 0;
            const int leader = __ffsll((long long)m) - 1;
            uint32_t avail = 0;
            if constexpr (DENSE) {
                const uint32_t cnt = (uint32_t)__popcll(m);
                avail = res_end - res_next;
                if (cnt > avail) {
                    if (lane == leader) first = atomicAdd(queue, cnt - avail + kReserve);
                    first = __shfl(first, leader, 64);
                }
            } else {
                if (lane == leader) first = atomicAdd(queue, (uint32_t)__popcll(m));
                first = __shfl(first, leader, 64);
            }
            if (need) {
                uint32_t i = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                if constexpr (DENSE) {
                    const uint32_t r = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    i = r < avail ? res_next + r : first + (r - avail);
                }
                if (i < nitems) {
                    if constexpr (!Source::kRing) src.get(i, base, len, dst);
                    blk = 0;
                    nblk = (len + 8) / 64 + 1;
                    have = true;
                } else {
                    exhausted = true;
                    have = false;
                }
            }
            if constexpr (DENSE) {
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (cnt > avail) {
                    res_next = first + (cnt - avail);
                    res_end = res_next + kReserve;
                } else {
                    res_next += cnt;
                }
            }
        };
        // request the lane's next block into FIFO slot s (compile-time s)
        auto prep = [&](const int s) {
            acquire(!have && !exhausted);
            uint32_t c = 0;
            const uint8_t *p = reinterpret_cast<const uint8_t *>(kIdleBlock);
            const uint8_t *bb = base;
            uint64_t off = 0;
            bool tail = false, last = false;
            if (have) {
                off = blk * 64;
                if constexpr (Source::kRing) bb = (off < len1) ? base : base2;  // which physical page holds this block
                last = blk + 1 == nblk;
                if (off + 64 <= len) p = bb + off;  // pure data block: 4-byte aligned vector loads + funnel selector
                else tail = true;                   // tail / padding block (<= 2 per range)
                c = 1u | (last ? 2u : 0u);
                dstv[s] = dst;
                if constexpr (Source::kRing) pagesv[s] = pages;
                if (++blk == nblk) have = false;
            }
            sha256_request_block(p, R[s], selv[s]);
            if (tail) {
                sha256_tail_words(bb, len, off, last, R[s]);
                selv[s] = 0x00010203u;
            }
            cflag[s] = c;
        };

#pragma unroll
        for (int s = 0; s < D; ++s) prep(s);
        // ONE exit, behind the last slot's step. Every path from a slot's requests back to the perm that consumes them then
        // passes the other slot's five requests, and the slot is awaited with vmcnt(5). (An exit in the middle — a test of
        // `running` per slot, or a return — is routed through the loop's latch by the structuriser: an edge from slot 0's
        // barrier to the loop's head on which nothing follows slot 0's requests, and the wait at the head is vmcnt(0).)
        // If the pairs drain at slot 0's barrier, slot 1's step is one empty step whose barrier only the producer waves
        // still reach: the consumers have ended, and s_barrier waits for the surviving waves of a workgroup only.
        bool drained = false;
        [[maybe_unused]] bool wg_busy = false;  // some pair of this workgroup had a block in the previous step
        [[maybe_unused]] RingProbe probe;
        [[maybe_unused]] const bool probe_on = blockIdx.x == 0 && wave == 2;
        while (!drained) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                {
                    constexpr int kBufMask = 1;
                    const int pb = s & kBufMask;  // D is even: buffer parity is static
                    uint32_t W[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        W[j] = __builtin_amdgcn_perm(R[s][j + 1], R[s][j], selv[s]);
                        // pinned HERE (an empty volatile statement is neither sunk into the `any_cur` branch below nor
                        // moved behind prep(s)): the slot's registers must be dead before its refill is requested, or
                        // the refill lands in a second register set that is copied back at the loop's end behind an
                        // s_waitcnt vmcnt(0) — every block would again be awaited in the step that requested it
                        asm volatile("" : "+v"(W[j]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t c = cflag[s];
                    uint8_t *cur_dst = dstv[s];
                    if constexpr (Source::kRing) {
                        // The chunk's LAST block has arrived in registers: nothing of the chunk will be read from HBM
                        // again. Drop its page references (release: all earlier loads of this lane have completed);
                        // whoever brings a page to zero hands it back to the host, which may refill it at once.
                        // (Round 4 tried letting go of a two-page chunk's FIRST page as soon as its last block there was
                        // in: no gain — a max-size chunk lives almost entirely in ONE 16.2 MiB page, which it holds for
                        // its whole 0.46 s either way; configs[2] through the ring 412 vs 422 GiB/s. Removed again.)
                        if (c & 2u) {
                            const uint32_t pg = pagesv[s];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const uint32_t pi = h ? (pg >> 16) : (pg & 0xffffu);
                                if (pi != 0xffffu) {
                                    const uint32_t old = __hip_atomic_fetch_sub(&src.pending[pi], 1u, __ATOMIC_RELEASE,
                                                                                __HIP_MEMORY_SCOPE_AGENT);
                                    if (old == 1u) {
                                        const uint32_t fs = atomicAdd(&src.ctl->free_count, 1u);
                                        __hip_atomic_store(&src.free_fifo[fs & src.free_mask],
                                                           ((unsigned long long)(fs + 1u) << 32) | pi, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_SYSTEM);
                                    }
                                }
                            }
                        }
                    }
                    prep(s);  // refill the slot: the block D iterations ahead
                    const bool any_cur = __any(c & 1u);
                    if (any_cur) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            uint32_t x[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int t = 4 * q + j;
                                x[j] = ((t < 16) ? W[t] : sha256_sched(W, t)) + kSha256K[t];
                            }
                            wkbuf[pb][q][lane] = make_uint4(x[0], x[1], x[2], x[3]);
                        }
                    }
                    ctrl[pb][lane] = c;
                    dstp[pb][lane] = cur_dst;
                    bool live = any_cur;
                    if constexpr (Source::kRing) {
                        // a service wave stays until `stop` has reached every lane and its block FIFO has drained; with
                        // nothing to do it naps instead of spinning through barriers.
                        // A host that died, or sits in a blocking read for longer than idle_ticks, must not leave a kernel
                        // behind that never ends: the DESIGNATED wave (workgroup 0, first producer) then stops the service on
                        // its own. That must never lose a chunk, i.e. no round may publish positions behind lanes that have
                        // left. Handshake (Dekker, through mapped pinned memory): the wave announces its intent, THEN re-reads
                        // the heartbeat and the host's count of enqueued rounds; the host bumps heartbeat and count FIRST and
                        // THEN looks at the intent flag before it enqueues a round (ring.cpp, ring_service_gate). Either the
                        // wave sees the host and withdraws, or the host sees the intent and waits for the outcome. The stop
                        // is committed only if every enqueued round has been published (rounds_done) — published positions
                        // are always served: a lane leaves only at a position the tail has not reached.
                        const bool wave_done = __ballot(!exhausted) == 0ull;
                        live = any_cur || !wave_done;
                        if (!any_cur && !wave_done) {
                            const unsigned long long now = wall_clock64();
                            if (idle_since == 0) idle_since = now;
                            if (blockIdx.x == 0 && wave == 2 && now - idle_since > src.idle_ticks) {
                                // (the heartbeat lives in HOST memory: it is read once per timeout, not per idle iteration —
                                // hundreds of idle waves polling it over PCIe every few microseconds slowed the ring down 6x)
                                uint32_t *hbw = const_cast<uint32_t *>(src.heartbeat);
                                const uint32_t hb = __hip_atomic_load(hbw + kHbBeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                if (hb != hb_seen) {  // alive: it just has nothing for us yet
                                    hb_seen = hb;
                                    idle_since = now;
                                } else {
                                    if (lane == 0) __hip_atomic_store(hbw + kHbIntent, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    __atomic_thread_fence(__ATOMIC_SEQ_CST);  // system scope: the intent is out before the re-reads
                                    __threadfence_system();
                                    const uint32_t hb2 = __hip_atomic_load(hbw + kHbBeat, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    const uint32_t enq = __hip_atomic_load(hbw + kHbRoundsEnq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    const uint32_t done = __hip_atomic_load(&src.ctl->rounds_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    if (hb2 != hb || enq != done) {  // the host is back, or a round is still on its way: withdraw
                                        if (lane == 0) __hip_atomic_store(hbw + kHbIntent, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                        hb_seen = hb2;
                                        idle_since = now;
                                    } else if (lane == 0) {  // commit: every lane leaves at its next look at {tail, stop}
                                        __hip_atomic_store(&src.ctl->stop, 2u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                                        __hip_atomic_store(hbw + kHbCommitted, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                                    }
                                }
                            }
                            // (nap only while the WHOLE workgroup is idle: the pairs of a workgroup share the block barrier, and
                            // an idle pair that slept 1.3 us per step held its working sibling at 1.9 us per block instead of 1.7
                            // — the chain of every chunk in a draining ring, of every chunk of a lone file)
                            if (!wg_busy) __builtin_amdgcn_s_sleep(48);
                        } else {
                            idle_since = 0;
                        }
                        if (lane == 0) curw[pr][pb] = any_cur ? 1u : 0u;
                        if (probe_on) ring_probe_step(src, probe, any_cur, 0, lane, (uint32_t)__popcll(__ballot(c & 1u)));
                    }
                    if (lane == 0) alive[pr][pb] = live ? 1u : 0u;
                    __syncthreads();
                    drained = drained || __builtin_amdgcn_readfirstlane(any_alive(pb)) == 0u;  // every pair drained
                    if constexpr (Source::kRing) {
                        uint32_t w = curw[0][pb] | curw[1][pb];
                        if constexpr (DENSE) w |= curw[2][pb] | curw[3][pb];
                        wg_busy = w != 0u;
                    }
                }
            }
        }
    } else {
        // CONSUMER: software-pipelined over blocks. After barrier k+1 (producer has published block k+1) it
        // first issues the LDS reads of block k+1 into the OTHER register set and only then runs the 64 rounds
        // of block k from registers, so LDS latency and barrier skew hide behind ~3700 cycles of rounds. The
        // early copy to registers is also what keeps two LDS buffers sufficient.
        uint32_t H[8];
        sha256_iv(H);
        struct Blk {
            uint32_t al, c;
            uint8_t *d;
            uint4 wk[16];
        };
        auto fetch = [&](Blk &x, const int pb) {
            x.al = any_alive(pb);
            x.c = ctrl[pb][lane];
            x.d = dstp[pb][lane];
#pragma unroll
            for (int q = 0; q < 16; ++q) x.wk[q] = wkbuf[pb][q][lane];
        };
        auto rounds = [&](const Blk &x) {
            if (x.c & 1u) {
                uint32_t a = H[0], b = H[1], cc = H[2], dd = H[3], e = H[4], f = H[5], g = H[6], h = H[7];
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    SHA256_ROUND(a, b, cc, dd, e, f, g, h, x.wk[q].x);
                    SHA256_ROUND(h, a, b, cc, dd, e, f, g, x.wk[q].y);
                    SHA256_ROUND(g, h, a, b, cc, dd, e, f, x.wk[q].z);
                    SHA256_ROUND(f, g, h, a, b, cc, dd, e, x.wk[q].w);
                    SHA256_ROUND(e, f, g, h, a, b, cc, dd, x.wk[q + 1].x);
                    SHA256_ROUND(dd, e, f, g, h, a, b, cc, x.wk[q + 1].y);
                    SHA256_ROUND(cc, dd, e, f, g, h, a, b, x.wk[q + 1].z);
                    SHA256_ROUND(b, cc, dd, e, f, g, h, a, x.wk[q + 1].w);
                }
                H[0] += a; H[1] += b; H[2] += cc; H[3] += dd; H[4] += e; H[5] += f; H[6] += g; H[7] += h;
                if (x.c & 2u) {
                    // (a GLOBAL pointer as well: one flat_store anywhere in the kernel — the two roles share one function — leaves
                    // a "flat access may be pending" state on the paths into the producer's loop, and with it every wait
                    // for a block slot becomes vmcnt(0) lgkmcnt(0))
                    PBSK_GLOBAL uint32_t *o = (PBSK_GLOBAL uint32_t *)x.d;
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(H[j]);
                    if constexpr (Source::kRing) {  // record cell in mapped pinned memory: digest first, then its flag
                        __threadfence_system();
                        __hip_atomic_store(o + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
                    sha256_iv(H);
                }
            }
        };
        Blk A, B;
        __syncthreads();  // barrier 0: block 0 is published
        fetch(A, 0);
        for (;;) {
            if (!A.al) break;          // block k is the drained marker: the producers have left after its barrier
            __syncthreads();           // barrier k+1
            fetch(B, 1);
            __builtin_amdgcn_sched_barrier(0);
            rounds(A);
            if (!B.al) break;
            __syncthreads();           // barrier k+2
            fetch(A, 0);
            __builtin_amdgcn_sched_barrier(0);
            rounds(B);
        }
    }
}

// -------------------------------------------------------------------------------------
// SHA-256, EXPRESS form: TWO LANES PER CHUNK, on top of the wave pair. Rounds 1-3 called the chain of a max-size chunk a
// floor: 64 rounds x 14 instructions, one instruction per ~4.2 cycles per wave whatever its lane count, and two WAVES
// cannot share one chain (an LDS round trip >> the 60 cycles of a round). Two LANES of the same wave can: DPP operands
// move a value to the neighbouring lane inside the instruction that uses it. A round's two halves
//     e' = Sigma1(e) + Ch(e,f,g) + [h + d + K + W]        (needs the a-chain only through d = a three rounds ago)
//     a' = Sigma0(a) + Maj(a,b,c) + [e' - d]               (needs the e-chain only through e')
// are the SAME nine instructions on different operands:
//     lane A (even) keeps e,f,g,h = X0..X3, lane B (odd) keeps a,b,c,d and runs TWO rounds behind A, so that ONE
//     register name serves both directions of the exchange: A reads d = a(r-3) from B's X1, B reads e(r-1) from A's X1;
//     Sigma: three v_alignbit by per-lane shift registers (6,11,25 | 2,13,22) + xor3;
//     Ch | Maj: bfi(X0 ^ (X2 & ROLE), X1, X2) — ROLE = 0 gives Ch(e,f,g), ROLE = ~0 gives bfi(a^c, b, c) = Maj(a,b,c);
//     the bracket: NZ = (X3 ^ ROLE) + C (v_xad_u32: h + [K+W] for A with C = K+W; -d for B with C = 1), then
//     P = partner's X1 + NZ (v_add_u32_dpp quad_perm:[1,0,3,2]); X0' = S + F + P (v_add3).
// 9 instructions per round instead of 14, 66 slot-rounds per block (B's two rounds of lag) + 12 for the per-lane
// selects at the block's ends: ~630 issue slots per block against ~946 — the serial chain of a chunk runs 1.5x faster
// (oracle: the formulation was first checked lane by lane in Python against hashlib, scripts/r4_xpair_sim.py).
// The PRODUCER wave needs no cross-lane work at all: the message schedule does not depend on the state, so its two lanes
// of a chunk expand two CONSECUTIVE blocks at once (lane A block 2m, lane B block 2m+1, each with the ordinary code) and
// a producer step (one barrier) feeds two consumer blocks: plane E and plane O of the LDS buffer, both in the row of lane
// A; the rows of the B lanes hold the constant 1 (C above), written once.
// Cost: a wave carries 32 chunks instead of 64, i.e. 0.74 of the pair form's throughput per CU — the form is for the
// chunks whose chain bounds something (the ring's long-chunk queue, a lone file), not for all of them.
// Four slot-rounds in ONE block of fixed order (36 instructions; the start variant adds the two selects that put b and a
// into B's chain). One block, because (1) the DPP operand of a round (its X1) must have been written at least two
// instructions before the v_add_u32_dpp that reads it and the hazard recogniser does not look into inline assembly: X1 is
// an input of the block or the result of a round at least nine instructions earlier; (2) the compiler pads every boundary
// between two assembly statements with an s_nop, and an s_nop costs a whole issue slot of the lone wave.
#define XP_R(n, x0, x1, x2, x3, kw)                                                     \
    "v_xad_u32 %[nz], " x3 ", %[role], " kw "\n\t"                                      \
    "v_alignbit_b32 %[r1], " x0 ", " x0 ", %[sh1]\n\t"                                  \
    "v_alignbit_b32 %[r2], " x0 ", " x0 ", %[sh2]\n\t"                                  \
    "v_add_u32_dpp %[p], " x1 ", %[nz] quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t" \
    "v_alignbit_b32 %[r3], " x0 ", " x0 ", %[sh3]\n\t"                                  \
    "v_bitop3_b32 %[sel], " x0 ", " x2 ", %[role] bitop3:0x78\n\t"                      \
    "v_bitop3_b32 %[sg], %[r1], %[r2], %[r3] bitop3:0x96\n\t"                           \
    "v_bitop3_b32 %[f], %[sel], " x1 ", " x2 " bitop3:0xca\n\t"                         \
    "v_add3_u32 " n ", %[sg], %[f], %[p]\n\t"
#define XP_TEMPS                                                                                                   \
    [nz] "=&v"(nz), [r1] "=&v"(r1), [r2] "=&v"(r2), [r3] "=&v"(r3), [p] "=&v"(p), [sel] "=&v"(sel), [sg] "=&v"(sg), \
        [f] "=&v"(f)
struct XpRole {
    uint32_t sh1, sh2, sh3, role;
};
// rounds r .. r+3 of a block; (x0..x3) in: the chain's last four values, newest first; out: the same after four rounds.
// START: r = 0 — B's results of slot-rounds 0 and 1 are replaced by hb1 (= b) and hb0 (= a): its chain starts two late.
template <bool START>
__device__ __forceinline__ void xp_round4(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3, const uint4 kw, const XpRole &R,
                                          const uint32_t hb1 = 0, const uint32_t hb0 = 0) {
    uint32_t n0, n1, n2, n3, nz, r1, r2, r3, p, sel, sg, f;
    if constexpr (START) {
        asm(XP_R("%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[k0]")
            "v_bitop3_b32 %[n0], %[role], %[hb1], %[n0] bitop3:0xca\n\t"
            XP_R("%[n1]", "%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[k1]")
            "v_bitop3_b32 %[n1], %[role], %[hb0], %[n1] bitop3:0xca\n\t"
            XP_R("%[n2]", "%[n1]", "%[n0]", "%[x0]", "%[x1]", "%[k2]")
            XP_R("%[n3]", "%[n2]", "%[n1]", "%[n0]", "%[x0]", "%[k3]")
            : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), XP_TEMPS
            : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [k0] "v"(kw.x), [k1] "v"(kw.y), [k2] "v"(kw.z), [k3] "v"(kw.w),
              [sh1] "v"(R.sh1), [sh2] "v"(R.sh2), [sh3] "v"(R.sh3), [role] "v"(R.role), [hb1] "v"(hb1), [hb0] "v"(hb0));
    } else {
        asm(XP_R("%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[k0]")
            XP_R("%[n1]", "%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[k1]")
            XP_R("%[n2]", "%[n1]", "%[n0]", "%[x0]", "%[x1]", "%[k2]")
            XP_R("%[n3]", "%[n2]", "%[n1]", "%[n0]", "%[x0]", "%[k3]")
            : [n0] "=&v"(n0), [n1] "=&v"(n1), [n2] "=&v"(n2), [n3] "=&v"(n3), XP_TEMPS
            : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [k0] "v"(kw.x), [k1] "v"(kw.y), [k2] "v"(kw.z), [k3] "v"(kw.w),
              [sh1] "v"(R.sh1), [sh2] "v"(R.sh2), [sh3] "v"(R.sh3), [role] "v"(R.role));
    }
    x0 = n3; x1 = n2; x2 = n1; x3 = n0;
}
// the two slot-rounds behind round 63 (B finishes; A idles): K+W is irrelevant to B (its constant is the row of ones)
__device__ __forceinline__ void xp_round2(uint32_t &x0, uint32_t &x1, uint32_t &x2, uint32_t &x3, const uint32_t kw, const XpRole &R) {
    uint32_t n0, n1, nz, r1, r2, r3, p, sel, sg, f;
    asm(XP_R("%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[x3]", "%[k0]")
        XP_R("%[n1]", "%[n0]", "%[x0]", "%[x1]", "%[x2]", "%[k0]")
        : [n0] "=&v"(n0), [n1] "=&v"(n1), XP_TEMPS
        : [x0] "v"(x0), [x1] "v"(x1), [x2] "v"(x2), [x3] "v"(x3), [k0] "v"(kw), [sh1] "v"(R.sh1), [sh2] "v"(R.sh2), [sh3] "v"(R.sh3),
          [role] "v"(R.role));
    x3 = x1; x2 = x0; x1 = n0; x0 = n1;
}
// role ? b : a, bitwise (ROLE is 0 or ~0 per lane)
__device__ __forceinline__ uint32_t xp_sel(uint32_t role, uint32_t a, uint32_t b) { return __builtin_amdgcn_bitop3_b32(role, b, a, 0xCA); }

template <typename Source>
__global__ __launch_bounds__(256) void k_sha256_xpair(Source src, const uint32_t *nitems_p, uint32_t nitems_imm, uint32_t *queue,
                                                      const uint32_t *wg_limit) {
    if (wg_limit && *wg_limit == 0) return;  // k_order: skip this pass (a scan tile overflowed, the batch is re-run)
    // waves 0,1 = consumers of pair 0,1; waves 2,3 = their producers. 32 chunks per pair: chunk ci = lanes 2ci (A), 2ci+1 (B).
    __shared__ uint4 wkbuf_[2][2][2][16][64];  // [pair][buffer][plane E/O][4 rounds][row]: 128 KiB
    __shared__ uint32_t ctrl_[2][2][2][32];    // [pair][buffer][plane][chunk]: bit0 block valid, bit1 last block of its chunk
    __shared__ uint8_t *dstp_[2][2][2][32];    // digest destination (valid when bit1)
    __shared__ uint32_t alive[2][2];           // [pair][buffer]: producer still had blocks
    __shared__ uint32_t curw[2][2];            // [pair][buffer]: the pair had a block in this step
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int pr = wave & 1;
    const bool producer = wave >= 2;
    const bool roleB = (lane & 1) != 0;
    const int ci = lane >> 1;
    auto &wkbuf = wkbuf_[pr];
    auto &ctrl = ctrl_[pr];
    auto &dstp = dstp_[pr];
    auto any_alive = [&](const int pb) -> uint32_t { return alive[0][pb] | alive[1][pb]; };
    // rows of the B lanes: the constant 1 in every plane of every buffer (never written again)
    for (int i = threadIdx.x; i < 2 * 2 * 2 * 16 * 32; i += 256) {
        const int row = 2 * (i & 31) + 1, q = (i >> 5) & 15, pl = (i >> 9) & 1, pb = (i >> 10) & 1, pp = (i >> 11) & 1;
        wkbuf_[pp][pb][pl][q][row] = make_uint4(1u, 1u, 1u, 1u);
    }
    __syncthreads();

    if (producer) {
        const uint32_t nitems = nitems_p ? *nitems_p : nitems_imm;
        const uint8_t *base = nullptr;
        uint64_t len = 0, blk = 0, nblk = 0;  // blk = the lane's next block (A: even blocks, B: odd blocks)
        uint8_t *dst = nullptr;
        bool have = false, exhausted = false;
        [[maybe_unused]] const uint8_t *base2 = nullptr;
        [[maybe_unused]] uint32_t len1 = 0, pages = 0xffffffffu;
        [[maybe_unused]] unsigned long long idle_since = 0;
        [[maybe_unused]] uint32_t poll_ctr = 0;
        constexpr int D = 2;
        uint32_t R[D][17];
        uint32_t selv[D], cflag[D];
        uint8_t *dstv[D];
        [[maybe_unused]] uint32_t pagesv[D];
#pragma unroll
        for (int s = 0; s < D; ++s) {
#pragma unroll
            for (int j = 0; j < 17; ++j) R[s][j] = 0;
            selv[s] = 0x00010203u;
            cflag[s] = 0;
            dstv[s] = nullptr;
            pagesv[s] = 0xffffffffu;
        }
        // `need`: this A lane's PAIR has issued every block of its chunk and wants the next one (B lanes never ask)
        auto acquire = [&](bool need) {
            if (__ballot(need) == 0) return;
            bool got = false;
            if constexpr (Source::kRing) {
                // the ring's LONG-chunk queue only (RingSource::ldesc): no claims, no waiting at positions — the wave's
                // leader moves lhead forward by compare-and-swap over published positions
                if (__ballot(have) != 0 && ((++poll_ctr) & src.poll_mask) != 0u) return;
                const unsigned long long lq = __hip_atomic_load(&src.ctl->lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t lt = (uint32_t)lq, lh = (uint32_t)(lq >> 32);
                const int32_t avail = (int32_t)(lt - lh);
                const unsigned long long mn = __ballot(need);
                if (avail > 0) {
                    const uint32_t cnt = min((uint32_t)__popcll(mn), (uint32_t)avail);
                    const int leader = __ffsll((long long)mn) - 1;
                    uint32_t got0 = 0xffffffffu;
                    if (lane == leader && atomicCAS(&src.ctl->lhead, lh, lh + cnt) == lh) got0 = lh;
                    got0 = __shfl(got0, leader, 64);
                    if (got0 != 0xffffffffu) {
                        __atomic_thread_fence(__ATOMIC_ACQUIRE);
                        const uint32_t rank = (uint32_t)__popcll(mn & ((1ull << lane) - 1ull));
                        got = need && rank < cnt;
                        uint4 e0 = make_uint4(0, 0, 0, 0), e1 = make_uint4(0, 0, 0, 0);
                        if (got) {
                            e0 = src.ldesc[2u * ((got0 + rank) & src.lmask)];
                            e1 = src.ldesc[2u * ((got0 + rank) & src.lmask) + 1u];
                        }
                        base = got ? reinterpret_cast<const uint8_t *>(((uint64_t)e0.y << 32) | e0.x) : base;
                        base2 = got ? reinterpret_cast<const uint8_t *>(((uint64_t)e1.y << 32) | e1.x) : base2;
                        len = got ? (uint64_t)e0.z : len;
                        len1 = got ? e0.w : len1;
                        dst = got ? src.cells + (uint64_t)e1.z * 64u + 8u : dst;
                        pages = got ? e1.w : pages;
                        if (lane == leader) atomicAdd(&src.ctl->xp_busy, cnt);  // (given back when the chunk's last block is in)
                    }
                } else {
                    // nothing published: has the service been told to stop? (stop is raised behind the last publish; the
                    // queue is looked at again behind the fence, so no lane leaves while long chunks are unclaimed)
                    const unsigned long long ts = __hip_atomic_load(&src.ctl->tail_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((uint32_t)(ts >> 32) != 0u) {
                        __atomic_thread_fence(__ATOMIC_ACQUIRE);
                        const unsigned long long lq2 = __hip_atomic_load(&src.ctl->lq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if ((int32_t)((uint32_t)lq2 - (uint32_t)(lq2 >> 32)) <= 0) exhausted = exhausted | need;
                    }
                }
            } else {
                const unsigned long long m = __ballot(need);
                uint32_t first = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) first = atomicAdd(queue, (uint32_t)__popcll(m));
                first = __shfl(first, leader, 64);
                if (need) {
                    const uint32_t i = first + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    if (i < nitems) {
                        src.get(i, base, len, dst);
                        got = true;
                    } else {
                        exhausted = true;
                    }
                }
            }
            // the chunk (or the end of work) goes to the pair's B lane
            const int srcl = lane & ~1;
            const bool pgot = __shfl((int)got, srcl, 64) != 0;
            exhausted = __shfl((int)exhausted, srcl, 64) != 0;
            if (__ballot(pgot)) {
                const uint64_t b1 = (uint64_t)__shfl((unsigned long long)reinterpret_cast<uintptr_t>(base), srcl, 64);
                const uint64_t ln = (uint64_t)__shfl((unsigned long long)len, srcl, 64);
                const uint64_t ds = (uint64_t)__shfl((unsigned long long)reinterpret_cast<uintptr_t>(dst), srcl, 64);
                if (pgot) {
                    base = reinterpret_cast<const uint8_t *>(b1);
                    len = ln;
                    dst = reinterpret_cast<uint8_t *>(ds);
                }
                if constexpr (Source::kRing) {
                    const uint64_t b2 = (uint64_t)__shfl((unsigned long long)reinterpret_cast<uintptr_t>(base2), srcl, 64);
                    const uint32_t l1 = (uint32_t)__shfl((int)len1, srcl, 64);
                    const uint32_t pg = (uint32_t)__shfl((int)pages, srcl, 64);
                    if (pgot) {
                        base2 = reinterpret_cast<const uint8_t *>(b2);
                        len1 = l1;
                        pages = pg;
                    }
                }
                if (pgot) {
                    nblk = (len + 8) / 64 + 1;
                    blk = roleB ? 1ull : 0ull;
                    have = blk < nblk;
                }
            }
        };
        auto prep = [&](const int s) {
            const unsigned long long hm = __ballot(have);
            const bool pair_busy = ((hm >> (lane & ~1)) & 3ull) != 0ull;
            acquire(!roleB && !pair_busy && !exhausted);
            uint32_t c = 0;
            const uint8_t *p = reinterpret_cast<const uint8_t *>(kIdleBlock);  // (unconditional request: k_sha256_pair)
            const uint8_t *bb = base;
            uint64_t off = 0;
            bool tail = false, last = false;
            if (have) {
                off = blk * 64;
                if constexpr (Source::kRing) bb = (off < len1) ? base : base2;
                last = blk + 1 == nblk;
                if (off + 64 <= len) p = bb + off;
                else tail = true;
                c = 1u | (last ? 2u : 0u);
                dstv[s] = dst;
                if constexpr (Source::kRing) pagesv[s] = pages;
                blk += 2;
                if (blk >= nblk) have = false;
            }
            sha256_request_block(p, R[s], selv[s]);
            if (tail) {
                sha256_tail_words(bb, len, off, last, R[s]);
                selv[s] = 0x00010203u;
            }
            cflag[s] = c;
        };
        // where this lane's expanded block goes: plane E (A lanes) or O (B lanes), always the row of the pair's A lane
        const int wrow = lane & ~1, wplane = lane & 1;

#pragma unroll
        for (int s = 0; s < D; ++s) prep(s);
        bool drained = false;  // (one exit behind the last slot: k_sha256_pair)
        [[maybe_unused]] bool wg_busy = false;
        [[maybe_unused]] RingProbe probe;
        [[maybe_unused]] const bool probe_on = blockIdx.x == 0 && wave == 2;
        while (!drained) {
#pragma unroll
            for (int s = 0; s < D; ++s) {
                {
                    const int pb = s & 1;
                    uint32_t W[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        W[j] = __builtin_amdgcn_perm(R[s][j + 1], R[s][j], selv[s]);
                        // pinned HERE (an empty volatile statement is neither sunk into the `any_cur` branch below nor
                        // moved behind prep(s)): the slot's registers must be dead before its refill is requested, or
                        // the refill lands in a second register set that is copied back at the loop's end behind an
                        // s_waitcnt vmcnt(0) — every block would again be awaited in the step that requested it
                        asm volatile("" : "+v"(W[j]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const uint32_t c = cflag[s];
                    uint8_t *cur_dst = dstv[s];
                    if constexpr (Source::kRing) {
                        // the chunk's LAST block is in registers (the partner's blocks of this step and all earlier ones
                        // too: same load instructions, same wait): the pair is free for the service's accounting, and the
                        // chunk's page references are dropped, as the pair form does
                        {
                            const unsigned long long mdone = __ballot((c & 2u) != 0u);
                            if (mdone && lane == (__ffsll((long long)mdone) - 1)) atomicSub(&src.ctl->xp_busy, (uint32_t)__popcll(mdone));
                        }
                        if (c & 2u) {
                            const uint32_t pg = pagesv[s];
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                const uint32_t pi = h ? (pg >> 16) : (pg & 0xffffu);
                                if (pi != 0xffffu) {
                                    const uint32_t old = __hip_atomic_fetch_sub(&src.pending[pi], 1u, __ATOMIC_RELEASE,
                                                                                __HIP_MEMORY_SCOPE_AGENT);
                                    if (old == 1u) {
                                        const uint32_t fs = atomicAdd(&src.ctl->free_count, 1u);
                                        __hip_atomic_store(&src.free_fifo[fs & src.free_mask],
                                                           ((unsigned long long)(fs + 1u) << 32) | pi, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_SYSTEM);
                                    }
                                }
                            }
                        }
                    }
                    prep(s);
                    const bool any_cur = __any(c & 1u);
                    if (any_cur) {
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            uint32_t x[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int t = 4 * q + j;
                                x[j] = ((t < 16) ? W[t] : sha256_sched(W, t)) + kSha256K[t];
                            }
                            wkbuf[pb][wplane][q][wrow] = make_uint4(x[0], x[1], x[2], x[3]);
                        }
                    }
                    ctrl[pb][wplane][ci] = c;
                    dstp[pb][wplane][ci] = cur_dst;
                    bool live = any_cur;
                    if constexpr (Source::kRing) {
                        // stays until `stop` has reached every pair and the block FIFO has drained; naps while idle. (The
                        // idle self-stop of a silent host is the pair service's business: it raises `stop` for both.)
                        const bool wave_done = __ballot(!exhausted) == 0ull;
                        live = any_cur || !wave_done;
                        if (!any_cur && !wave_done && !wg_busy) __builtin_amdgcn_s_sleep(48);  // (never while the sibling pair works)
                        if (lane == 0) curw[pr][pb] = any_cur ? 1u : 0u;
                        if (probe_on) ring_probe_step(src, probe, any_cur, 3, lane);  // (a step of this form = TWO blocks per chunk)
                    }
                    if (lane == 0) alive[pr][pb] = live ? 1u : 0u;
                    __syncthreads();
                    drained = drained || __builtin_amdgcn_readfirstlane(any_alive(pb)) == 0u;
                    if constexpr (Source::kRing) wg_busy = (curw[0][pb] | curw[1][pb]) != 0u;
                }
            }
        }
    } else {
        // CONSUMER: lane A = e,f,g,h (H4..H7), lane B = a,b,c,d (H0..H3), lock-step
        const XpRole R{roleB ? 2u : 6u, roleB ? 13u : 11u, roleB ? 22u : 25u, roleB ? 0xffffffffu : 0u};
        const uint32_t role = R.role;
        uint32_t ivr[4];
        ivr[0] = roleB ? 0x6a09e667u : 0x510e527fu;
        ivr[1] = roleB ? 0xbb67ae85u : 0x9b05688cu;
        ivr[2] = roleB ? 0x3c6ef372u : 0x1f83d9abu;
        ivr[3] = roleB ? 0xa54ff53au : 0x5be0cd19u;
        uint32_t HR[4] = {ivr[0], ivr[1], ivr[2], ivr[3]};
        struct Blk {
            uint32_t al, c;
            uint8_t *d;
            uint4 wk[16];
        };
        auto fetch = [&](Blk &x, const int pb, const int plane) {
            x.al = any_alive(pb);
            x.c = ctrl[pb][plane][ci];
            x.d = dstp[pb][plane][ci];
#pragma unroll
            for (int q = 0; q < 16; ++q) x.wk[q] = wkbuf[pb][plane][q][lane];
        };
        auto rounds = [&](const Blk &x) {
            if (x.c & 1u) {
                // slot-round r: A computes e(r+1), B computes a(r-1). B's first two slot-rounds only shift its start
                // values into place (a(-1) = b, a(0) = a), A's last two are idle: per-lane selects at both ends.
                uint32_t x0 = xp_sel(role, HR[0], HR[2]), x1 = xp_sel(role, HR[1], HR[3]), x2 = HR[2], x3 = HR[3];
                xp_round4<true>(x0, x1, x2, x3, x.wk[0], R, HR[1], HR[0]);
#pragma unroll
                for (int g = 1; g < 16; ++g) xp_round4<false>(x0, x1, x2, x3, x.wk[g], R);
                // after slot-round 63: (x0..x3) = results of slot-rounds 63, 62, 61, 60 = A's e(64..61)
                const uint32_t o63 = x0, o62 = x1, o61 = x2, o60 = x3;
                xp_round2(x0, x1, x2, x3, x.wk[15].w, R);
                // ... after 65: x0, x1 = results of 65, 64; B's a(64..61) = results of 65, 64, 63, 62
                HR[0] += xp_sel(role, o63, x0);
                HR[1] += xp_sel(role, o62, x1);
                HR[2] += xp_sel(role, o61, o63);
                HR[3] += xp_sel(role, o60, o62);
                if (x.c & 2u) {
                    // B holds digest words 0..3, A words 4..7: one 16-byte store each
                    PBSK_GLOBAL uint32_t *o = (PBSK_GLOBAL uint32_t *)x.d + (roleB ? 0 : 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) o[j] = __builtin_bswap32(HR[j]);
                    if constexpr (Source::kRing) {  // record cell in mapped pinned memory: both halves first, then the flag
                        __threadfence_system();
                        __hip_atomic_store((PBSK_GLOBAL uint32_t *)x.d + 10, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    }
#pragma unroll
                    for (int j = 0; j < 4; ++j) HR[j] = ivr[j];
                }
            }
        };
        // Software-pipelined like the pair form: the LDS reads of the NEXT block are issued before the rounds of the current
        // one run from registers. A producer step = two blocks (planes E, O) and one barrier.
        Blk E0, O0, E1, O1;
        __syncthreads();  // barrier 0: step 0 is published (buffer 0)
        fetch(E0, 0, 0);
        for (;;) {
            if (!E0.al) break;
            fetch(O0, 0, 1);
            __builtin_amdgcn_sched_barrier(0);
            rounds(E0);
            __syncthreads();  // step k+1 published (buffer 1); the producer may now overwrite buffer 0: O0 is in registers
            fetch(E1, 1, 0);
            __builtin_amdgcn_sched_barrier(0);
            rounds(O0);
            if (!E1.al) break;
            fetch(O1, 1, 1);
            __builtin_amdgcn_sched_barrier(0);
            rounds(E1);
            __syncthreads();
            fetch(E0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            rounds(O1);
        }
    }
}

// -------------------------------------------------------------------------------------
// Longest-first queue order + workgroup budget for the SHA kernel (one small workgroup).
// A batch's makespan is max(longest chunk, total work / lanes): starting the long chunks first
// and letting lanes pull the short ones afterwards reaches that bound with FEWER lanes than
// chunks, which leaves CUs free for the next batch's kernels (batches overlap on separate
// streams). Counting sort by size class (no comparison sort needed for a scheduling order).
__global__ __launch_bounds__(1024) void k_order(const uint8_t *data, const pbsgpu_segment *segs, const pbsgpu_record *recs,
                                                const uint32_t *nrec_p, uint32_t shift, uint4 *qdesc, uint32_t *wg_limit,
                                                uint32_t max_wgs,
                                                const uint32_t *maxcnt, uint32_t cap, uint32_t slack_pct) {
    // a scan tile overflowed its slot list: this pass will be re-run with a larger capacity, so do not
    // spend a SHA pass on its (incomplete) cut list
    if (maxcnt && *maxcnt > cap) {
        if (threadIdx.x == 0) *wg_limit = 0;
        return;
    }
    constexpr int BINS = 1024;
    __shared__ uint32_t hist[BINS];
    __shared__ uint32_t base[BINS];
    __shared__ unsigned long long tot_blocks;
    __shared__ uint32_t longest;
    const uint32_t n = *nrec_p;
    for (int i = threadIdx.x; i < BINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) { tot_blocks = 0; longest = 0; }
    __syncthreads();
    unsigned long long my_blocks = 0;
    uint32_t my_long = 0;
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t sz = recs[i].size;
        uint32_t b = sz >> shift;
        if (b >= BINS) b = BINS - 1;
        atomicAdd(&hist[BINS - 1 - b], 1u);  // descending size
        const uint32_t blocks = (sz + 8) / 64 + 1;
        my_blocks += blocks;
        my_long = max(my_long, blocks);
    }
    atomicAdd(&tot_blocks, my_blocks);
    atomicMax(&longest, my_long);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < BINS; ++i) { base[i] = run; run += hist[i]; }
        // lanes needed so that total work / lanes stays below the longest chain (slack_pct, default 25 %)
        const unsigned long long lg = longest ? longest : 1;
        unsigned long long lanes = (tot_blocks * (100 + slack_pct) / 100 + lg - 1) / lg;
        uint32_t wgs = (uint32_t)((lanes + 127) / 128);
        const uint32_t need = (n + 127) / 128;
        if (wgs > need) wgs = need;
        if (wgs < 1) wgs = 1;
        if (wgs > max_wgs) wgs = max_wgs;
        *wg_limit = wgs;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        uint32_t b = recs[i].size >> shift;
        if (b >= BINS) b = BINS - 1;
        const uint32_t pos = atomicAdd(&base[BINS - 1 - b], 1u);
        const uint64_t p = (uint64_t)(uintptr_t)(data + segs[recs[i].segment].offset + recs[i].end - recs[i].size);
        qdesc[pos] = make_uint4((uint32_t)p, (uint32_t)(p >> 32), recs[i].size, i);
    }
}

// dense_pct: work per pair-mode lane, in percent of the longest chain, from which the SHA kernel runs its dense (4 pairs
// per CU) form; 0 = never (pbsgpu_engine_options::sha_dense_pct, default 150)
bool sha256_dense_pays(uint64_t total_blocks, uint64_t longest_blocks, int num_cus, uint32_t dense_pct) {
    if (!dense_pct) return false;
    if (longest_blocks < 1) longest_blocks = 1;
    return total_blocks * 100ull > (uint64_t)dense_pct * longest_blocks * 128ull * (uint64_t)num_cus;
}

hipError_t launch_order(const uint8_t *data, const pbsgpu_segment *segs, const pbsgpu_record *recs, const uint32_t *nrec,
                        uint32_t max_chunk, uint4 *qdesc, uint32_t *wg_limit, int num_cus, const uint32_t *maxcnt, uint32_t cap, uint32_t slack_pct,
                        hipStream_t st) {
    uint32_t shift = 0;
    while (((uint64_t)max_chunk >> shift) >= 1024) ++shift;
    hipLaunchKernelGGL(k_order, dim3(1), dim3(1024), 0, st, data, segs, recs, nrec, shift, qdesc, wg_limit, (uint32_t)num_cus,
                       maxcnt, cap, slack_pct);
    return hipGetLastError();
}

// Dynamic-LDS padding: a wave already saturates its SIMD's integer issue rate (one wave64 VALU op
// per ~4 cycles), so a co-resident wave halves the speed of the serial chain. Requesting LDS the
// kernel never touches caps residency at one wave per SIMD (2 pairs / 4 single waves per CU).
constexpr size_t kPairLdsPad = 16u << 10;  // ~69 KB static + 16 KB > 80 KB -> exactly one pair workgroup per CU
constexpr size_t kLaneLdsPad = 36u << 10;  // four single-wave workgroups per CU

template <typename K>
static hipError_t allow_lds(K kernel, size_t bytes) {
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               (int)bytes);
}

// `form` (pbsgpu_engine_options::sha_form): 0 = wave pairs (default), 1 = the single-wave kernel, 2 = the express form (two
// lanes per chunk) for EVERY chunk of the batch path — A/B measurements and the parity tests of those kernels
// host-decided form (descriptor jobs, whole-segment hashing): exactly one launch
template <typename Source>
static hipError_t launch_pair(unsigned grid, bool dense, hipStream_t st, Source src, uint32_t nitems, uint32_t *queue, int form) {
    if (form == 2) {  // (131 KB static LDS: one workgroup per CU without padding)
        hipLaunchKernelGGL((k_sha256_xpair<Source>), dim3(grid * (dense ? 4u : 2u)), dim3(256), 0, st, src, (const uint32_t *)nullptr,
                           nitems, queue, (const uint32_t *)nullptr);
        return hipGetLastError();
    }
    if (dense) {
        hipLaunchKernelGGL((k_sha256_pair<Source, true>), dim3(grid), dim3(512), 0, st, src, (const uint32_t *)nullptr,
                           nitems, queue, (const uint32_t *)nullptr);
    } else {
        const size_t pad = kPairLdsPad;
        hipError_t e = allow_lds(&k_sha256_pair<Source, false>, pad);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_sha256_pair<Source, false>), dim3(grid), dim3(256), pad, st, src,
                           (const uint32_t *)nullptr, nitems, queue, (const uint32_t *)nullptr);
    }
    return hipGetLastError();
}


hipError_t launch_sha256_records(pbsgpu_record *recs, const uint32_t *nrec, uint32_t *queue, const uint4 *qdesc,
                                 const uint32_t *wg_limit, int num_cus, bool dense, int form, hipStream_t st) {
    RecordSource src{recs, qdesc};
    if (form == 2) {
        hipLaunchKernelGGL((k_sha256_xpair<RecordSource>), dim3((unsigned)num_cus), dim3(256), 0, st, src, nrec, 0u, queue,
                           wg_limit);
    } else if (form == 0) {
        if (dense) {
            hipLaunchKernelGGL((k_sha256_pair<RecordSource, true>), dim3((unsigned)num_cus), dim3(512), 0, st, src, nrec,
                               0u, queue, wg_limit);
        } else {
            hipError_t e = allow_lds(&k_sha256_pair<RecordSource, false>, kPairLdsPad);
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL((k_sha256_pair<RecordSource, false>), dim3((unsigned)num_cus), dim3(256), kPairLdsPad, st, src,
                               nrec, 0u, queue, wg_limit);
        }
    } else {
        hipError_t e = allow_lds(&k_sha256<RecordSource>, kLaneLdsPad);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_sha256<RecordSource>), dim3((unsigned)num_cus * 4u), dim3(64), kLaneLdsPad, st, src, nrec, 0u,
                           queue, wg_limit, 0u);
    }
    return hipGetLastError();
}

hipError_t launch_sha256_segments(const uint8_t *data, const pbsgpu_segment *segs, uint32_t nseg, uint8_t *digests,
                                  uint32_t *queue, int num_cus, bool dense, int form, hipStream_t st) {
    if (nseg == 0) return hipSuccess;
    SegmentSource src{data, segs, digests};
    if (form != 1) {
        unsigned g2 = (unsigned)num_cus;
        const unsigned need2 = (nseg + (dense ? 255u : 127u)) / (dense ? 256u : 128u);
        if (g2 > need2) g2 = need2;
        return launch_pair(g2, dense, st, src, nseg, queue, form);
    }
    hipError_t e = allow_lds(&k_sha256<SegmentSource>, kLaneLdsPad);
    if (e != hipSuccess) return e;
    // `dense` with this form (sha_form 1 + sha_dense_pct): EIGHT single-wave workgroups per CU, two waves per SIMD — a measurement
    // (profiles/r06_sha_forms_full_lanes.log section 3): with two waves on a SIMD v_add_u32 / v_xor_b32 issue at twice the rate
    // of the three-operand forms (profiles/r01_ubench_opcode_issue_cost.log)
    const size_t pad = dense ? kLaneLdsPad / 2 : kLaneLdsPad;
    unsigned grid = (unsigned)num_cus * (dense ? 8u : 4u);
    const unsigned need = (nseg + 63) / 64;
    if (grid > need) grid = need;
    hipLaunchKernelGGL((k_sha256<SegmentSource>), dim3(grid), dim3(64), pad, st, src, (const uint32_t *)nullptr,
                       nseg, queue, (const uint32_t *)nullptr, 0u);
    return hipGetLastError();
}

// =====================================================================================
// small device -> host publication without the copy engines
// =====================================================================================
// hipMemcpyAsync(D2H) rides an SDMA queue that HIP streams share: a copy that has to wait for the kernel in front of
// it on ITS stream (e.g. "scalars after the 0.4 s SHA launch") parks at the head of that queue and every later copy
// of every other stream — another batch's segment-table upload, a stream window's record readback — waits behind it
// (measured: 380-430 ms stalls of a 3 KB readback while another stream's SHA kernel ran). Results that a host thread
// waits for are therefore WRITTEN by a kernel straight into mapped pinned host memory.
__global__ __launch_bounds__(256) void k_publish(uint32_t *dst, const uint32_t *src, uint64_t nwords) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride) dst[i] = src[i];
}

// count-dependent form: publishes recs[0 .. *nrec) (12 words each, capped at cap records)
__global__ __launch_bounds__(256) void k_publish_records(uint32_t *dst, const uint32_t *src, const uint32_t *nrec,
                                                         uint64_t cap) {
    const uint64_t n = min((uint64_t)*nrec, cap) * (sizeof(pbsgpu_record) / 4);
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}

hipError_t launch_publish(void *dst_host_mapped, const void *src, uint64_t nbytes, hipStream_t st) {
    if (nbytes == 0) return hipSuccess;
    const uint64_t nwords = (nbytes + 3) / 4;
    unsigned blocks = (unsigned)std::min<uint64_t>((nwords + 255) / 256, 64);
    hipLaunchKernelGGL(k_publish, dim3(blocks), dim3(256), 0, st, (uint32_t *)dst_host_mapped, (const uint32_t *)src, nwords);
    return hipGetLastError();
}

hipError_t launch_publish_records(pbsgpu_record *dst_host_mapped, const pbsgpu_record *src, const uint32_t *nrec,
                                  uint64_t cap, hipStream_t st) {
    if (cap == 0) return hipSuccess;
    unsigned blocks = (unsigned)std::min<uint64_t>((cap * 12 + 255) / 256, 64);
    hipLaunchKernelGGL(k_publish_records, dim3(blocks), dim3(256), 0, st, (uint32_t *)dst_host_mapped, (const uint32_t *)src,
                       nrec, cap);
    return hipGetLastError();
}

// =====================================================================================
// synthetic corpus generator (twin of oracle_fill)
// =====================================================================================
__device__ __forceinline__ uint64_t splitmix64(uint64_t seed, uint64_t idx) {
    uint64_t z = seed + (idx + 1) * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// kind 4: 16-byte blocks from two ChaCha quarter-rounds over {block index, seed}: adds / xors / rotates only (24 full-rate
// VALU ops per 16 bytes; splitmix64's two 64-bit multiplies cost ~6 ops per BYTE at quarter rate). The page ring's refill
// runs inside the timed region on the few CUs the SHA service leaves free, so the generator has to be cheap.
__device__ __forceinline__ uint4 chacha2_block(uint64_t q, uint64_t seed) {
    uint32_t x0 = (uint32_t)q ^ 0x61707865u, x1 = (uint32_t)(q >> 32) ^ 0x3320646eu;
    uint32_t x2 = (uint32_t)seed ^ 0x79622d32u, x3 = (uint32_t)(seed >> 32) ^ 0x6b206574u;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        x0 += x1; x3 = __builtin_rotateleft32(x3 ^ x0, 16);
        x2 += x3; x1 = __builtin_rotateleft32(x1 ^ x2, 12);
        x0 += x1; x3 = __builtin_rotateleft32(x3 ^ x0, 8);
        x2 += x3; x1 = __builtin_rotateleft32(x1 ^ x2, 7);
    }
    return make_uint4(x0, x1, x2, x3);
}

__device__ __forceinline__ uint64_t fill_word(uint64_t widx, uint64_t seed, uint32_t kind) {
    switch (kind) {
    case 0: return splitmix64(seed, widx);
    case 4: {
        const uint4 b = chacha2_block(widx >> 1, seed);
        return (widx & 1u) ? ((uint64_t)b.w << 32) | b.z : ((uint64_t)b.y << 32) | b.x;
    }
    case 1: return 0;
    case 2: return splitmix64(seed, widx & 511u);
    default: {
        const uint64_t g = widx >> 13;
        const uint64_t r = splitmix64(seed ^ 0xA5A5A5A55A5A5A5Aull, g);
        if ((((r >> 32) * 10u) >> 32) < 3u) return 0;
        return splitmix64(seed, widx);
    }
    }
}

__global__ __launch_bounds__(256) void k_fill(uint64_t *dst, uint64_t w0, uint64_t nwords, uint64_t seed,
                                              uint32_t kind, uint8_t *tail_dst, uint32_t tail_bytes) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nwords; i += stride)
        dst[i] = fill_word(w0 + i, seed, kind);
    if (blockIdx.x == 0 && threadIdx.x == 0 && tail_bytes) {
        const uint64_t w = fill_word(w0 + nwords, seed, kind);
        for (uint32_t b = 0; b < tail_bytes; ++b) tail_dst[b] = (uint8_t)(w >> (8 * b));
    }
}

hipError_t launch_fill(void *dptr, uint64_t stream_off, uint64_t nbytes, uint64_t seed, uint32_t kind,
                       hipStream_t st) {
    if (nbytes == 0) return hipSuccess;
    const uint64_t nwords = nbytes / 8;
    const uint32_t tail = (uint32_t)(nbytes & 7u);
    uint64_t blocks = (nwords + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_fill, dim3((unsigned)blocks), dim3(256), 0, st, (uint64_t *)dptr, stream_off >> 3, nwords,
                       seed, kind, (uint8_t *)dptr + nwords * 8, tail);
    return hipGetLastError();
}

// =====================================================================================
// digest-set duplicate detection (cross-file dedup over the all-gathered record set)
// =====================================================================================
// Sort (first 8 digest bytes, index) pairs with a stable radix sort, then a record is a
// duplicate iff an earlier entry of its equal-prefix run carries the same 32-byte digest.
__global__ __launch_bounds__(256) void k_dedup_keys(const pbsgpu_record *recs, uint64_t n, uint64_t *keys,
                                                    uint32_t *idx) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint8_t *d = recs[i].digest;
    uint64_t k = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) k = (k << 8) | d[b];
    keys[i] = k;
    idx[i] = (uint32_t)i;
}

__device__ __forceinline__ bool digest_eq(const pbsgpu_record *a, const pbsgpu_record *b) {
    const uint32_t *x = reinterpret_cast<const uint32_t *>(a->digest);
    const uint32_t *y = reinterpret_cast<const uint32_t *>(b->digest);
    bool eq = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) eq &= (x[i] == y[i]);
    return eq;
}

__global__ __launch_bounds__(256) void k_dedup_mark(const pbsgpu_record *recs, uint64_t n, const uint64_t *keys,
                                                    const uint32_t *idx, uint8_t *dup, uint64_t *stats4) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    bool is_dup = false;
    uint64_t size = 0;
    if (j < n) {
        const pbsgpu_record *me = recs + idx[j];
        size = me->size;
        const uint64_t key = keys[j];
        for (uint64_t q = j; q > 0 && keys[q - 1] == key; --q) {
            if (digest_eq(me, recs + idx[q - 1])) {  // stable sort: idx[q-1] < idx[j]
                is_dup = true;
                break;
            }
        }
        if (dup) dup[idx[j]] = is_dup ? 1 : 0;
    }
    // stats: [0] records, [1] unique, [2] total bytes, [3] unique bytes (wave-reduced atomics)
    uint64_t c_all = (j < n) ? 1 : 0, c_uni = (j < n && !is_dup) ? 1 : 0;
    uint64_t b_all = size, b_uni = is_dup ? 0 : size;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        c_all += __shfl_xor(c_all, d, 64);
        c_uni += __shfl_xor(c_uni, d, 64);
        b_all += __shfl_xor(b_all, d, 64);
        b_uni += __shfl_xor(b_uni, d, 64);
    }
    if ((threadIdx.x & 63) == 0 && c_all) {
        atomicAdd(reinterpret_cast<unsigned long long *>(stats4 + 0), (unsigned long long)c_all);
        atomicAdd(reinterpret_cast<unsigned long long *>(stats4 + 1), (unsigned long long)c_uni);
        atomicAdd(reinterpret_cast<unsigned long long *>(stats4 + 2), (unsigned long long)b_all);
        atomicAdd(reinterpret_cast<unsigned long long *>(stats4 + 3), (unsigned long long)b_uni);
    }
}

size_t dedup_tmp_bytes(uint64_t n) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, (uint64_t *)nullptr, (uint64_t *)nullptr, (uint32_t *)nullptr,
                                    (uint32_t *)nullptr, (size_t)n, 0, 64, (hipStream_t)0);
    return bytes + 256;
}

hipError_t launch_dedup(const pbsgpu_record *recs, uint64_t n, uint64_t *keys, uint32_t *idx, uint64_t *keys_alt,
                        uint32_t *idx_alt, uint8_t *dup, uint64_t *stats4, void *tmp, size_t tmp_bytes,
                        hipStream_t st) {
    hipError_t e = hipMemsetAsync(stats4, 0, 4 * sizeof(uint64_t), st);
    if (e != hipSuccess || n == 0) return e;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_dedup_keys, dim3(nb), dim3(256), 0, st, recs, n, keys, idx);
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, keys, keys_alt, idx, idx_alt, (size_t)n, 0, 64, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_dedup_mark, dim3(nb), dim3(256), 0, st, recs, n, keys_alt, idx_alt, dup, stats4);
    return hipGetLastError();
}

// =====================================================================================
// payload-stream assembly (pxar v2 split archive, .ppxar): [start marker] { [header 16 B][content] }* [tail]
// =====================================================================================
// One work item = one <= 4 MiB piece of one file (table built on the host, which knows the file
// list). A workgroup copies its piece with 16-byte stores to the (arbitrarily aligned) destination;
// sources are read as 4-byte-aligned dwords and funnel-shifted (v_alignbyte) into place.
__device__ __forceinline__ uint32_t load_shifted(const uint8_t *src) {  // 4 bytes at any alignment
    const uint32_t o = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t *a = reinterpret_cast<const uint32_t *>(src - o);
    const uint32_t lo = a[0];
    if (o == 0) return lo;
    return __builtin_amdgcn_alignbyte(a[1], lo, o);
}

__global__ __launch_bounds__(256) void k_pack(const uint8_t *src_base, uint8_t *dst, const PackItem *items,
                                              uint32_t nitems) {
    const uint32_t it = blockIdx.x;
    if (it >= nitems) return;
    const PackItem w = items[it];
    uint8_t *d = dst + w.dst_off;
    if (w.kind != 0) {  // 16-byte header / marker: {type u64 LE, size u64 LE}
        if (threadIdx.x < 16) {
            const uint64_t v = (threadIdx.x < 8) ? w.src_off /* type */ : w.len /* size field */;
            d[threadIdx.x] = (uint8_t)(v >> (8 * (threadIdx.x & 7)));
        }
        return;
    }
    const uint8_t *sp = src_base + w.src_off;
    const uint64_t n = w.len;
    // head: bytes until the destination is 16-byte aligned
    uint64_t head = (16 - ((uintptr_t)d & 15u)) & 15u;
    if (head > n) head = n;
    if (threadIdx.x < head) d[threadIdx.x] = sp[threadIdx.x];
    const uint64_t body = (n - head) / 16;
    const uint8_t *sb = sp + head;
    uint8_t *db = d + head;
    for (uint64_t i = threadIdx.x; i < body; i += blockDim.x) {
        const uint8_t *q = sb + i * 16;
        uint4 v;
        if ((((uintptr_t)q) & 3u) == 0) {
            const u32x4_a4 t = *reinterpret_cast<const u32x4_a4 *>(q);
            v = make_uint4(t.x, t.y, t.z, t.w);
        } else {
            v = make_uint4(load_shifted(q), load_shifted(q + 4), load_shifted(q + 8), load_shifted(q + 12));
        }
        *reinterpret_cast<uint4 *>(db + i * 16) = v;
    }
    const uint64_t done = head + body * 16;
    const uint64_t tail = n - done;
    if (threadIdx.x < tail) d[done + threadIdx.x] = sp[done + threadIdx.x];
}

hipError_t launch_pack(const uint8_t *src_base, uint8_t *dst, const PackItem *items, uint32_t nitems, hipStream_t st) {
    if (nitems == 0) return hipSuccess;
    hipLaunchKernelGGL(k_pack, dim3(nitems), dim3(256), 0, st, src_base, dst, items, nitems);
    return hipGetLastError();
}

// =====================================================================================
// XXH3-64 (seed 0, default secret) of many byte ranges — the per-file hash the reference tees
// new file bodies through (xxh3.New(), internal/pxarmount/commit_reuse.go:450-461) and re-checks
// after the commit (commit_orchestrate.go:485-562).
// =====================================================================================
// One wave per byte range (see xxh::blocks below); inputs <= 240 bytes take the scalar formulas.
__device__ constexpr uint8_t kXxhSecret[192] = {
    0xb8, 0xfe, 0x6c, 0x39, 0x23, 0xa4, 0x4b, 0xbe, 0x7c, 0x01, 0x81, 0x2c, 0xf7, 0x21, 0xad, 0x1c, 0xde, 0xd4, 0x6d, 0xe9,
    0x83, 0x90, 0x97, 0xdb, 0x72, 0x40, 0xa4, 0xa4, 0xb7, 0xb3, 0x67, 0x1f, 0xcb, 0x79, 0xe6, 0x4e, 0xcc, 0xc0, 0xe5, 0x78,
    0x82, 0x5a, 0xd0, 0x7d, 0xcc, 0xff, 0x72, 0x21, 0xb8, 0x08, 0x46, 0x74, 0xf7, 0x43, 0x24, 0x8e, 0xe0, 0x35, 0x90, 0xe6,
    0x81, 0x3a, 0x26, 0x4c, 0x3c, 0x28, 0x52, 0xbb, 0x91, 0xc3, 0x00, 0xcb, 0x88, 0xd0, 0x65, 0x8b, 0x1b, 0x53, 0x2e, 0xa3,
    0x71, 0x64, 0x48, 0x97, 0xa2, 0x0d, 0xf9, 0x4e, 0x38, 0x19, 0xef, 0x46, 0xa9, 0xde, 0xac, 0xd8, 0xa8, 0xfa, 0x76, 0x3f,
    0xe3, 0x9c, 0x34, 0x3f, 0xf9, 0xdc, 0xbb, 0xc7, 0xc7, 0x0b, 0x4f, 0x1d, 0x8a, 0x51, 0xe0, 0x4b, 0xcd, 0xb4, 0x59, 0x31,
    0xc8, 0x9f, 0x7e, 0xc9, 0xd9, 0x78, 0x73, 0x64, 0xea, 0xc5, 0xac, 0x83, 0x34, 0xd3, 0xeb, 0xc3, 0xc5, 0x81, 0xa0, 0xff,
    0xfa, 0x13, 0x63, 0xeb, 0x17, 0x0d, 0xdd, 0x51, 0xb7, 0xf0, 0xda, 0x49, 0xd3, 0x16, 0x55, 0x26, 0x29, 0xd4, 0x68, 0x9e,
    0x2b, 0x16, 0xbe, 0x58, 0x7d, 0x47, 0xa1, 0xfc, 0x8f, 0xf8, 0xb8, 0xd1, 0x7a, 0xd0, 0x31, 0xce, 0x45, 0xcb, 0x3a, 0x8f,
    0x95, 0x16, 0x04, 0x28, 0xaf, 0xd7, 0xfb, 0xca, 0xbb, 0x4b, 0x40, 0x7e};

namespace xxh {
constexpr uint64_t P32_1 = 0x9E3779B1ull, P32_2 = 0x85EBCA77ull, P32_3 = 0xC2B2AE3Dull;
constexpr uint64_t P64_1 = 0x9E3779B185EBCA87ull, P64_2 = 0xC2B2AE3D27D4EB4Full, P64_3 = 0x165667B19E3779F9ull;
constexpr uint64_t P64_4 = 0x85EBCA77C2B2AE63ull, P64_5 = 0x27D4EB2F165667C5ull;
constexpr uint64_t PMX1 = 0x165667919E3779F9ull, PMX2 = 0x9FB21C651E98DF25ull;

__device__ __forceinline__ uint64_t sec64(int off) {  // little-endian 8 bytes of the secret (any offset)
    uint64_t v = 0;
#pragma unroll
    for (int b = 0; b < 8; ++b) v |= (uint64_t)kXxhSecret[off + b] << (8 * b);
    return v;
}
__device__ __forceinline__ uint64_t sec64_dyn(int off) {
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) v |= (uint64_t)kXxhSecret[off + b] << (8 * b);
    return v;
}
__device__ __forceinline__ uint64_t rd64(const uint8_t *p) {  // byte-safe little-endian load
    uint64_t v = 0;
    for (int b = 0; b < 8; ++b) v |= (uint64_t)p[b] << (8 * b);
    return v;
}
__device__ __forceinline__ uint32_t rd32(const uint8_t *p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }
__device__ __forceinline__ uint64_t fold128(uint64_t a, uint64_t b) { return (a * b) ^ __umul64hi(a, b); }
__device__ __forceinline__ uint64_t avalanche(uint64_t h) {
    h ^= h >> 37;
    h *= PMX1;
    return h ^ (h >> 32);
}
__device__ __forceinline__ uint64_t avalanche64(uint64_t h) {
    h ^= h >> 33;
    h *= P64_2;
    h ^= h >> 29;
    h *= P64_3;
    return h ^ (h >> 32);
}
__device__ __forceinline__ uint64_t mix16(const uint8_t *d, int so) {
    return fold128(rd64(d) ^ sec64_dyn(so), rd64(d + 8) ^ sec64_dyn(so + 8));
}
// inputs of 0..240 bytes (one lane)
__device__ uint64_t short_hash(const uint8_t *d, uint64_t n) {
    if (n == 0) return avalanche64(sec64(56) ^ sec64(64));
    if (n <= 3) {
        const uint32_t comb = ((uint32_t)d[0] << 16) | ((uint32_t)d[n >> 1] << 24) | (uint32_t)d[n - 1] | ((uint32_t)n << 8);
        const uint32_t flip = (uint32_t)sec64(0) ^ (uint32_t)(sec64(0) >> 32);
        return avalanche64((uint64_t)(comb ^ flip));
    }
    if (n <= 8) {
        const uint64_t in1 = rd32(d), in2 = rd32(d + n - 4);
        uint64_t h = (in2 + (in1 << 32)) ^ (sec64(8) ^ sec64(16));
        h ^= rotl64(h, 49) ^ rotl64(h, 24);
        h *= PMX2;
        h ^= (h >> 35) + n;
        h *= PMX2;
        return h ^ (h >> 28);
    }
    if (n <= 16) {
        const uint64_t lo = rd64(d) ^ (sec64(24) ^ sec64(32));
        const uint64_t hi = rd64(d + n - 8) ^ (sec64(40) ^ sec64(48));
        return avalanche(n + __builtin_bswap64(lo) + hi + fold128(lo, hi));
    }
    if (n <= 128) {
        uint64_t acc = n * P64_1;
        if (n > 32) {
            if (n > 64) {
                if (n > 96) acc += mix16(d + 48, 96) + mix16(d + n - 64, 112);
                acc += mix16(d + 32, 64) + mix16(d + n - 48, 80);
            }
            acc += mix16(d + 16, 32) + mix16(d + n - 32, 48);
        }
        acc += mix16(d, 0) + mix16(d + n - 16, 16);
        return avalanche(acc);
    }
    uint64_t acc = n * P64_1;
    for (int i = 0; i < 8; ++i) acc += mix16(d + 16 * i, 16 * i);
    acc = avalanche(acc);
    const int rounds = (int)(n / 16);
    for (int i = 8; i < rounds; ++i) acc += mix16(d + 16 * i, 16 * (i - 8) + 3);
    acc += mix16(d + n - 16, 136 - 17);
    return avalanche(acc);
}
}  // namespace xxh

// ---- long inputs (> 240 bytes): one WAVE per byte range ---------------------------------------------------
// XXH3's 512-bit state is 8 u64 accumulators; inside a 1 KiB block (16 stripes of 64 bytes) the update is a SUM
//   acc[i] += data64[i ^ 1] + lo32(data64[i] ^ key[s][i]) * hi32(data64[i] ^ key[s][i])
// and only the per-block scramble is non-linear. So all 64 lanes take part: lane = (stripe-in-half-block s8 =
// lane >> 3, accumulator i = lane & 7) adds its two stripes of the block (two fully coalesced 512-byte wave loads),
// the 8 partial sums per accumulator are reduced with three butterfly steps, and every lane applies the scramble to
// its (replicated) accumulator. Loads are byte-aligned 8-byte loads (gfx950 global loads are alignment-free).
namespace xxh {

// streaming state of ONE open byte range (the stream writer's per-file tee); hist sits directly in front of pend so
// that the final stripe (the last 64 bytes of the whole input) can be read at pend + pend_len - 64 even when fewer
// than 64 bytes are pending
struct State {
    uint64_t acc[8];
    uint64_t total;      // bytes seen so far
    uint32_t pend_len;   // bytes waiting in pend (1..1024 once anything was seen)
    uint32_t started;
    uint8_t hist[64];
    uint8_t pend[1024];
};

struct Keys {
    uint64_t k0, k1;     // stripe keys of this lane for the two half-blocks: secret[8 * (s8 + 8h) + 8 i ..]
    uint64_t scr, last, merge;
};

__device__ __forceinline__ Keys make_keys(int lane) {
    const int i = lane & 7, s8 = lane >> 3;
    Keys k;
    k.k0 = sec64_dyn(8 * s8 + 8 * i);
    k.k1 = sec64_dyn(8 * (s8 + 8) + 8 * i);
    k.scr = sec64_dyn(128 + 8 * i);
    k.last = sec64_dyn(121 + 8 * i);
    k.merge = sec64_dyn(11 + 8 * i);
    return k;
}

__device__ __forceinline__ uint64_t ld64(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

__device__ __forceinline__ uint64_t stripe_term(uint64_t v, uint64_t key) {
    const uint64_t nb = __shfl_xor(v, 1, 64);  // the neighbour accumulator's data word
    const uint64_t k = v ^ key;
    return nb + (uint64_t)(uint32_t)k * (uint64_t)(uint32_t)(k >> 32);
}

__device__ __forceinline__ uint64_t reduce_stripes(uint64_t part) {  // sum over the 8 lanes that share `i`
    part += __shfl_xor(part, 8, 64);
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    return part;
}

// Phase A — block sums. S_b[i] = sum over the 16 stripes of block b of the accumulate term; it does not depend on the
// accumulators, so every block of every input is an independent unit of work for any wave of the grid (perfect load
// balance whatever the file-size mix). Returned in all lanes (replicated over the 8 lanes with equal i).
__device__ __forceinline__ uint64_t block_sum(const uint8_t *blk, const Keys &k, int lane) {
    const uint8_t *q = blk + 8 * lane;  // 64 * s8 + 8 * i == 8 * lane
    return reduce_stripes(stripe_term(ld64(q), k.k0) + stripe_term(ld64(q + 512), k.k1));
}

// Phase B — the only serial part of a long input: acc = scramble(acc + S_b), block after block (~10 instructions each).
// S entries are 8 u64 per block; the next entries are in flight while one is applied.
__device__ __forceinline__ void chain(uint64_t &acc, const uint64_t *S, uint64_t nblk, const Keys &k, int lane) {
    const uint64_t *q = S + (lane & 7);
    constexpr int U = 8;
    uint64_t b = 0;
    for (; b + U <= nblk; b += U) {
        uint64_t v[U];
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = q[(b + j) * 8];
#pragma unroll
        for (int j = 0; j < U; ++j) {
            acc += v[j];
            acc ^= acc >> 47;
            acc ^= k.scr;
            acc *= P32_1;
        }
    }
    for (; b < nblk; ++b) {
        acc += q[b * 8];
        acc ^= acc >> 47;
        acc ^= k.scr;
        acc *= P32_1;
    }
}

// the tail of a long input: `rem` = the R (1..1024) bytes behind the last full block, n = total length (> 240);
// the 64 bytes in front of rem must be readable when R < 64 (the input itself, or State::hist)
__device__ __forceinline__ uint64_t finish_long(uint64_t acc, const uint8_t *rem, uint32_t R, uint64_t n, const Keys &k,
                                                int lane) {
    const int i = lane & 7, s8 = lane >> 3;
    const uint32_t nstripes = (R - 1) / 64;  // full stripes that are NOT the last one
    uint64_t part = 0;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t st = (uint32_t)s8 + 8u * h;
        const bool on = st < nstripes;                       // octet-uniform
        const uint64_t v = on ? ld64(rem + 64 * st + 8 * i) : 0;
        const uint64_t t = stripe_term(v, h ? k.k1 : k.k0);  // shuffles stay wave-converged
        part += on ? t : 0;
    }
    acc += reduce_stripes(part);
    acc += stripe_term(ld64(rem + R - 64 + 8 * i), k.last);  // last stripe: the final 64 bytes of the input
    const uint64_t mine = acc ^ k.merge;
    const uint64_t other = __shfl_xor(mine, 1, 64);
    uint64_t fold = ((i & 1) == 0) ? fold128(mine, other) : 0;
    fold += __shfl_xor(fold, 2, 64);
    fold += __shfl_xor(fold, 4, 64);
    return avalanche(n * P64_1 + fold);
}

__device__ __forceinline__ uint64_t init_acc(int lane) {
    const uint64_t init[8] = {P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1};
    uint64_t a = init[0];
#pragma unroll
    for (int j = 1; j < 8; ++j) a = ((lane & 7) == j) ? init[j] : a;
    return a;
}

// whole input in one piece; S = its block sums (phase A)
__device__ __forceinline__ uint64_t one_shot(const uint8_t *d, uint64_t n, const uint64_t *S, const Keys &k, int lane) {
    if (n <= 240) return short_hash(d, n);  // every lane computes it (wave-uniform branch); cheap
    uint64_t acc = init_acc(lane);
    const uint64_t nblk = (n - 1) / 1024;
    chain(acc, S, nblk, k, lane);
    return finish_long(acc, d + nblk * 1024, (uint32_t)(n - nblk * 1024), n, k, lane);
}

// global memory written by some lanes of this wave is about to be read by others (or the reverse): complete the
// outstanding accesses first (workgroup-scope fences: one L1 per CU, no cache maintenance, just the waits)
__device__ __forceinline__ void mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void wave_copy(uint8_t *dst, const uint8_t *src, uint32_t n, int lane) {
    for (uint32_t j = lane; j < n; j += 64) dst[j] = src[j];
}

// one piece of an input that arrives in several pieces (stream windows). flags: 1 = first piece, 2 = last piece.
// Blocks are consumed only while at least one more byte is known to follow (XXH3 treats the final <= 1024 bytes
// specially), so between pieces 1..1024 bytes wait in State::pend. The HOST mirrors the pending length (pure
// arithmetic on the piece lengths) and passes it with the number of blocks to consume now (nproc); phase A has already
// completed the pending block in place (if any) and summed all nproc blocks into S.
__device__ __forceinline__ bool piece(State *st, const uint8_t *p, uint64_t L, uint32_t flags, uint32_t pend,
                                      uint64_t nproc, const uint64_t *S, const Keys &k, int lane, uint64_t *result) {
    const bool first = (flags & 1u) != 0;
    uint64_t acc = first ? init_acc(lane) : st->acc[lane & 7];
    const uint64_t before = first ? 0ull : st->total;
    const uint64_t T = (uint64_t)pend + L;
    if (nproc) {
        chain(acc, S, nproc, k, lane);
        const uint64_t used = nproc * 1024 - pend;  // bytes of p inside the consumed blocks
        // history = the 64 bytes in front of the new pending tail
        uint8_t hb;
        if (used >= 64) hb = p[used - 64 + lane];
        else hb = (lane < 64 - (int)used) ? st->pend[1024 - (64 - used) + lane] : p[lane - (64 - used)];
        mem_sync();  // the completed pending block has been read by every lane before it is overwritten
        st->hist[lane] = hb;
        const uint32_t rest = (uint32_t)(L - used);
        wave_copy(st->pend, p + used, rest, lane);
        pend = rest;
    } else if (L) {
        wave_copy(st->pend + pend, p, (uint32_t)L, lane);
        pend = (uint32_t)T;
    }
    const uint64_t total = before + L;
    if (lane < 8) st->acc[lane] = acc;
    if (lane == 0) { st->total = total; st->pend_len = pend; st->started = 1; }
    mem_sync();  // pend / hist written by other lanes are read below
    if (!(flags & 2u)) return false;
    if (total <= 240) *result = short_hash(st->pend, total);  // nothing was ever consumed: pend holds the whole input
    else *result = finish_long(acc, st->pend, pend, total, k, lane);
    return true;
}
}  // namespace xxh

// Work item of the XXH3 kernels: hash bytes [ptr, ptr + len). flags bit0 = first piece of its input, bit1 = last piece;
// both set = a whole input (no state). Pieces of one input must be issued in order on one stream. pend / nproc / s_off
// are filled by the host (xxh3_plan): pending bytes in front of this piece, full 1 KiB blocks to consume now, index of
// the item's first block-sum entry.
// Phase A: one unit = kRun consecutive blocks of the global block list (all inputs back to back).
constexpr uint32_t kXxhRun = 16;
__global__ __launch_bounds__(256) void k_xxh3_sums(const XxhItem *items, uint32_t nitems, uint64_t total_blocks,
                                                   xxh::State *states, uint64_t *S) {
    using namespace xxh;
    const int lane = threadIdx.x & 63;
    const Keys k = make_keys(lane);
    const uint64_t nunits = (total_blocks + kXxhRun - 1) / kXxhRun;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
    for (uint64_t u = wave; u < nunits; u += nwaves) {
        uint64_t g = u * kXxhRun;
        const uint64_t gend = min(g + kXxhRun, total_blocks);
        // item that owns block g: last item with s_off <= g (items are in s_off order; empty items share an offset)
        uint32_t lo = 0, hi = nitems;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (items[mid].s_off <= g) lo = mid; else hi = mid;
        }
        uint32_t it = lo;
        while (g < gend) {
            while (it + 1 < nitems && (items[it].nproc == 0 || g >= items[it].s_off + items[it].nproc)) ++it;
            const XxhItem item = items[it];
            const uint64_t j = g - item.s_off;  // block of this item
            const uint8_t *blk;
            if (item.pend) {
                State *st = states + item.state;
                const uint32_t take = 1024u - item.pend;
                if (j == 0) {  // complete the pending block in place (this wave is its only reader in this phase)
                    wave_copy(st->pend + item.pend, item.ptr, take, lane);
                    mem_sync();
                    blk = st->pend;
                } else {
                    blk = item.ptr + take + (j - 1) * 1024;
                }
            } else {
                blk = item.ptr + j * 1024;
            }
            const uint64_t sum = block_sum(blk, k, lane);
            if (lane < 8) S[g * 8 + lane] = sum;
            ++g;
        }
    }
}

// Phase B: one wave per item
__global__ __launch_bounds__(256) void k_xxh3(const XxhItem *items, uint32_t nitems, xxh::State *states, const uint64_t *S,
                                              uint64_t *out, uint32_t *queue) {
    using namespace xxh;
    const int lane = threadIdx.x & 63;
    const Keys k = make_keys(lane);
    for (;;) {
        uint32_t idx = 0;
        if (lane == 0) idx = atomicAdd(queue, 1u);
        idx = __shfl(idx, 0, 64);
        if (idx >= nitems) break;
        const XxhItem it = items[idx];
        uint64_t h = 0;
        bool done;
        if ((it.flags & 3u) == 3u) {
            h = one_shot(it.ptr, it.len, S + it.s_off * 8, k, lane);
            done = true;
        } else {
            done = piece(states + it.state, it.ptr, it.len, it.flags, it.pend, it.nproc, S + it.s_off * 8, k, lane, &h);
        }
        if (done && lane == 0) out[it.out] = h;
    }
}

// fills pend-independent plan fields of whole inputs and returns the total number of blocks (host side)
uint64_t xxh3_plan_whole(XxhItem *items, uint32_t n) {
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        items[i].pend = 0;
        items[i].nproc = items[i].len > 240 ? (uint32_t)((items[i].len - 1) / 1024) : 0u;
        items[i].s_off = total;
        total += items[i].nproc;
    }
    return total;
}

size_t xxh3_state_bytes() { return sizeof(xxh::State); }

hipError_t launch_xxh3_items(const XxhItem *items, uint32_t nitems, uint64_t total_blocks, void *states, uint64_t *sums,
                             uint64_t *out, uint32_t *queue, int num_cus, hipStream_t st) {
    if (nitems == 0) return hipSuccess;
    if (total_blocks) {
        const uint64_t units = (total_blocks + kXxhRun - 1) / kXxhRun;
        unsigned grid = (unsigned)std::min<uint64_t>((units + 3) / 4, (uint64_t)num_cus * 8u);
        hipLaunchKernelGGL(k_xxh3_sums, dim3(grid), dim3(256), 0, st, items, nitems, total_blocks, (xxh::State *)states, sums);
    }
    unsigned grid = (unsigned)num_cus * 2u;  // 8 waves per CU
    const unsigned need = (nitems + 3) / 4;
    if (grid > need) grid = need;
    hipLaunchKernelGGL(k_xxh3, dim3(grid), dim3(256), 0, st, items, nitems, (xxh::State *)states, (const uint64_t *)sums, out,
                       queue);
    return hipGetLastError();
}

#include "ring_kernels.inc"

}  // namespace pbsk
