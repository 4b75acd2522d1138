// eincm_binning.hip.h — device-side staging of a batch of event windows (eincm_set_windows).
//
// The loaders of the reference hand over events in time order (src/dataloaders/mvsec_loader.py:272-295); the engine wants them
// binned by (window, 32x32 source tile) and cut into segments.  This is a counting sort done on the GPU:
//   k_bin_hist     per 4096-event block: validate + per-tile histogram in LDS            -> blockhist (nblk, ntiles)
//   k_bin_scan     per (window, tile): exclusive prefix over that window's blocks         -> blockoff in place, tilecount
//   k_bin_tilescan one workgroup: exclusive scans of tile counts and of segment counts    -> tilebase, itembase, n_items
//   k_bin_scatter  per block: position = tilebase + blockoff + LDS rank                   -> ev_xy, ev_t (binned)
//   k_items / k_seg_minmax                                                                -> segments with their time range
//   k_segsort                                                                             -> the order inside a segment (see there)
// The sort is STABLE: a block is one wave walking its 1024 events in input order, and events of a wave-step that share a tile are
// ranked by lane (k_bin_scatter), so a tile's events keep the order they were handed over in; k_segsort is a stable sort of a
// segment's events by source pixel.  Staging the same window twice therefore gives the same arrays bit for bit.  The per-thread sums
// of k_gather depend on that order, so this is what makes a re-staged window reproduce its gradient exactly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "eincm_kernels.hip.h"

namespace eincm {

constexpr int BIN_CHUNK = 1024;          // events per staging block
constexpr int BIN_NT = 64;               // one wave per block: its LDS traffic is ordered, which the stable ranking relies on
constexpr int BIN_TRIPS = BIN_CHUNK / BIN_NT;
constexpr int BIN_MAX_TILES = 12288;     // LDS histogram capacity (48 KiB); larger sensors use the host path

struct BinBlock { int32_t win, start, count, first_blk; };    // start: global index of the block's first event

// err[0] = smallest global index of an event outside the sensor (INT_MAX if none); err[1] = same for non-finite t
__global__ __launch_bounds__(BIN_NT) void k_bin_hist(Geom g, const BinBlock* __restrict__ blks, const int16_t* __restrict__ xs,
                                                  const int16_t* __restrict__ ys, const double* __restrict__ ts,
                                                  uint32_t* __restrict__ blockhist, int* __restrict__ err)
{
    extern __shared__ uint32_t hist[];
    const BinBlock bb = blks[blockIdx.x];
    // all of the block's events are fetched before the first one is used: the block is ONE wavefront walking BIN_TRIPS trips, and with
    // the loads inside the loop every trip waited for its own round trip to memory (24 us for 1024 events)
    uint32_t xy[BIN_TRIPS]; double tt[BIN_TRIPS];
#pragma unroll
    for (int k = 0; k < BIN_TRIPS; ++k) {
        const int i = k * BIN_NT + (int)threadIdx.x;
        xy[k] = 0u; tt[k] = 0.0;
        if (i < bb.count) { const int e = bb.start + i; xy[k] = (uint32_t)(uint16_t)xs[e] | ((uint32_t)(uint16_t)ys[e] << 16); tt[k] = ts[e]; }
    }
    for (int i = threadIdx.x; i < g.ntiles; i += BIN_NT) hist[i] = 0u;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < BIN_TRIPS; ++k) {
        const int i = k * BIN_NT + (int)threadIdx.x;
        if (i >= bb.count) continue;
        const int e = bb.start + i;
        const int x = (int16_t)(xy[k] & 0xffffu), y = (int16_t)(xy[k] >> 16);
        const double t = tt[k];
        if (x < 0 || x >= g.W || y < 0 || y >= g.H) { atomicMin(&err[0], e); continue; }
        if (!(t - t == 0.0)) { atomicMin(&err[1], e); continue; }
        atomicAdd(&hist[(y / TS) * g.tilesX + (x / TS)], 1u);
    }
    __syncthreads();
    uint32_t* out = blockhist + (size_t)blockIdx.x * g.ntiles;
    for (int i = threadIdx.x; i < g.ntiles; i += BIN_NT) out[i] = hist[i];
}

// thread per (window, tile); win_blk (B+1): first block of each window
__global__ void k_bin_scan(Geom g, const int32_t* __restrict__ win_blk, uint32_t* __restrict__ blockhist, int32_t* __restrict__ tilecount)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.ntiles) return;
    const int b = idx / g.ntiles, tile = idx % g.ntiles;
    uint32_t run = 0;
    // sixteen blocks per trip, all loads before the first store: one load -> store -> load chain per block made this pass 280 us at 10^6
    // events (977 blocks per window) and 2.8 ms at 10^7, a quarter of staging
    const int k0 = win_blk[b], k1 = win_blk[b + 1];
    for (int k = k0; k < k1; k += 16) {
        uint32_t v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = (k + j < k1) ? blockhist[(size_t)(k + j) * g.ntiles + tile] : 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (k + j < k1) { blockhist[(size_t)(k + j) * g.ntiles + tile] = run; run += v[j]; }
    }
    tilecount[idx] = (int32_t)run;
}

// one workgroup of 1024 threads: tilebase = exclusive scan of tilecount, itembase = exclusive scan of ceil(count/seg)
__global__ __launch_bounds__(1024) void k_bin_tilescan(int M, int seg, const int32_t* __restrict__ tilecount, int32_t* __restrict__ tilebase,
                                                        int32_t* __restrict__ itembase, int32_t* __restrict__ totals /* [n_events, n_items] */)
{
    __shared__ int64_t sA[1024];
    __shared__ int64_t sB[1024];
    const int t = threadIdx.x;
    const int per = (M + 1023) / 1024;
    const int lo = min(t * per, M), hi = min(lo + per, M);
    int64_t a = 0, b = 0;
    for (int i = lo; i < hi; ++i) { const int c = tilecount[i]; a += c; b += (c + seg - 1) / seg; }
    sA[t] = a; sB[t] = b;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan
        int64_t va = 0, vb = 0;
        if (t >= off) { va = sA[t - off]; vb = sB[t - off]; }
        __syncthreads();
        sA[t] += va; sB[t] += vb;
        __syncthreads();
    }
    int64_t ra = sA[t] - a, rb = sB[t] - b;               // exclusive prefix of this thread's chunk
    for (int i = lo; i < hi; ++i) {
        const int c = tilecount[i];
        tilebase[i] = (int32_t)ra; itembase[i] = (int32_t)rb;
        ra += c; rb += (c + seg - 1) / seg;
    }
    if (t == 1023) { totals[0] = (int32_t)sA[1023]; totals[1] = (int32_t)sB[1023]; }
}

__global__ __launch_bounds__(BIN_NT) void k_bin_scatter(Geom g, const BinBlock* __restrict__ blks, const int16_t* __restrict__ xs,
                                                         const int16_t* __restrict__ ys, const double* __restrict__ ts,
                                                         const uint32_t* __restrict__ blockoff, const int32_t* __restrict__ tilebase,
                                                         uint32_t* __restrict__ ev_xy, double* __restrict__ ev_t)
{
    extern __shared__ uint32_t cnt[];            // where this block's next event of each tile goes: tile base + block offset + placed so far
    const BinBlock bb = blks[blockIdx.x];
    const int lane = threadIdx.x;
    const uint32_t* off = blockoff + (size_t)blockIdx.x * g.ntiles;
    const int32_t* tb = tilebase + (size_t)bb.win * g.ntiles;
    for (int i = lane; i < g.ntiles; i += BIN_NT) cnt[i] = (uint32_t)tb[i] + off[i];       // (one coalesced pass instead of two gathers per event)
    __builtin_amdgcn_wave_barrier();
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t xy[BIN_TRIPS]; double tt[BIN_TRIPS];        // every trip's events up front (see k_bin_hist)
#pragma unroll
    for (int k = 0; k < BIN_TRIPS; ++k) {
        const int i = k * BIN_NT + lane;
        xy[k] = 0u; tt[k] = 0.0;
        if (i < bb.count) { const int e = bb.start + i; xy[k] = (uint32_t)(uint16_t)xs[e] | ((uint32_t)(uint16_t)ys[e] << 16); tt[k] = ts[e]; }
    }
    const int nbits = 32 - __clz(max(g.ntiles - 1, 1));      // bits of a tile index
#pragma unroll
    for (int k = 0; k < BIN_TRIPS; ++k) {
        if (k * BIN_NT >= bb.count) break;                   // wave-uniform
        const int i = k * BIN_NT + lane;
        const int x = (int16_t)(xy[k] & 0xffffu), y = (int16_t)(xy[k] >> 16);
        const double t = tt[k];
        int tile = -1;
        // events outside the sensor / with a non-finite time were reported by k_bin_hist and are not placed
        if (i < bb.count && x >= 0 && x < g.W && y >= 0 && y < g.H && (t - t == 0.0)) tile = (y / TS) * g.tilesX + (x / TS);
        // stable rank: the lanes that share a tile, found bit by bit (a match-any in nbits ballots instead of one round per distinct
        // tile of the wavefront: up to 64 rounds of LDS read-modify-write before), ordered by lane (= input order); the tile's running
        // count advances once per trip, by the group's lowest lane, after every lane has read it
        unsigned long long grp = __ballot(tile >= 0);
        for (int bit = 0; bit < nbits; ++bit) {
            const bool one = ((tile >> bit) & 1) != 0;
            const unsigned long long bal = __ballot(one);
            grp &= one ? bal : ~bal;
        }
        uint32_t slot = 0u, base = 0u;
        if (tile >= 0) { base = cnt[tile]; slot = base + (uint32_t)__popcll(grp & lt_mask); }
        __builtin_amdgcn_wave_barrier();
        if (tile >= 0 && (grp & lt_mask) == 0ull) cnt[tile] = base + (uint32_t)__popcll(grp);
        __builtin_amdgcn_wave_barrier();
        if (tile >= 0) {
            const size_t pos = slot;
            ev_xy[pos] = (uint32_t)(uint16_t)x | ((uint32_t)(uint16_t)y << 16);
            ev_t[pos] = t;
        }
    }
}

// Re-deal the events inside every block of 256 consecutive events of a tile (the 256 events the threads of an event-kernel workgroup
// take in one trip): sorted by source pixel (column-major), then dealt round-robin to the eight 32-event half-waves, so that the events a
// half-wave handles together come from different source COLUMNS (hence different pixels).  Events of one source pixel at nearby times land on one destination pixel,
// and LDS atomics are slower on shared destinations (profiles/r02/splat_bound_experiment.txt): k_splat gains 4-6 %.  The
// permutation is a function of the block's content alone (keys are unique: pixel, then position), so re-staging reproduces it;
// segments are multiples of 256 events from the tile's start, so no event changes segment.  A trailing partial block keeps its order.
// grid (B * ntiles, SPREAD_Y) workgroups of 256 threads: the blocks of a tile are dealt round-robin to SPREAD_Y workgroups.
constexpr int SPREAD_Y = 16;
__global__ __launch_bounds__(256) void k_spread(Geom g, const int32_t* __restrict__ tilecount, const int32_t* __restrict__ tilebase,
                                                uint32_t* __restrict__ ev_xy, double* __restrict__ ev_t)
{
    __shared__ __attribute__((aligned(16))) uint32_t keys[256];
    const int idx = blockIdx.x;
    const int cnt = tilecount[idx], base = tilebase[idx];
    const int t = threadIdx.x;
    for (int b0 = blockIdx.y * 256; b0 + 256 <= cnt; b0 += 256 * gridDim.y) {
        const uint32_t xy = ev_xy[base + b0 + t];
        const double tm = ev_t[base + b0 + t];
        // pixel inside the 32x32 tile (10 bits) above the position in the block (8 bits)
        // pixel inside the 32x32 tile, COLUMN-major (10 bits: x, then y), above the position in the block (8 bits)
        const uint32_t key = (((xy & 31u) << 5) | ((xy >> 16) & 31u)) << 8 | (uint32_t)t;
        __syncthreads();                                  // the previous block's keys are no longer read
        keys[t] = key;
        __syncthreads();
        int rank = 0;
        const uint4* k4 = reinterpret_cast<const uint4*>(keys);
#pragma unroll 8
        for (int k = 0; k < 64; ++k) {                    // broadcast reads, four keys each
            const uint4 q = k4[k];
            rank += (q.x < key ? 1 : 0) + (q.y < key ? 1 : 0) + (q.z < key ? 1 : 0) + (q.w < key ? 1 : 0);
        }
        // rank r of the column-major order goes to half-wave r mod 8, slot r / 8: the 32 lanes of a half-wave get ascending columns,
        // all different unless a column holds more than 8 of the block's events (win_pitch in eincm_kernels.hip.h: bank = column)
        const int hw = rank & 7;
        const int pos = (hw >> 1) * 64 + (hw & 1) * 32 + (rank >> 3);
        ev_xy[base + b0 + pos] = xy;                      // every thread has read its event: the block-local permutation is safe in place
        ev_t[base + b0 + pos] = tm;
    }
}

// k_segsort: the order of the events INSIDE a segment of the GATHER's copy of the events (round 3).  The two event kernels want
// different orders, so staging keeps two copies (12 B per event each): the splat's (time order inside a tile, re-dealt in blocks of
// 256: k_spread above - a wavefront holds events of nearly one time, so distinct source pixels go to distinct destination pixels) and
// this one.  A segment's events are sorted by source pixel (stable: input = time order), and the sorted sequence is dealt to the threads that
// will walk it so that a THREAD owns a contiguous run of it (SegWalk in eincm_kernels.hip.h): the segment is cut in two halves,
// a half of nc events is walked by 256 threads in K = ceil(nc / 256) coalesced steps, thread t taking sorted events
// [start_t, start_t + K or K - 1), event j of thread t stored at  half_base + j * 256 + t.  Consequences:
//   * consecutive events of a thread mostly share their source pixel: the theta-grid gather adds a RUN of -dt dL/dw in registers and
//     issues one pair of LDS atomics per run instead of per event, and re-reads the pixel's velocity only when the pixel changes;
//     the splat can merge the taps of consecutive events that round to the same destination;
//   The splat must NOT walk this order: its lanes would hold events of different times, and near the optimum events of one scene
//   point - different pixels at different times - warp onto one destination pixel (that is what contrast maximisation does), so a
//   wavefront's LDS atomics collide: k_splat 94 -> 123 us on the bench batch, 108 -> 172 us at 16x16 theta (profiles/r03/layout_experiments.md).
// The permutation is a function of the segment's content alone (stable counting sort, lanes ranked in lane order), so re-staging
// reproduces the arrays bit for bit.  Any order inside a segment is CORRECT (integer accumulation; the kernels walk every slot);
// this one is fast.  One workgroup of 4 waves per segment: each wave owns a quarter of the input (contiguous, in input order),
// counts its keys, and after the scan over (key, wave) places its quarter in order.
struct SegLayout {                 // where the sorted event of rank s (0 <= s < n) of a segment is stored, relative to the segment
    int n0, K0, K1, rem0, rem1;
    __host__ __device__ explicit SegLayout(int n) {
        n0 = (n + 1) >> 1;
        const int n1 = n - n0;
        K0 = (n0 + 255) >> 8; K1 = (n1 + 255) >> 8;
        rem0 = n0 - 256 * (K0 - 1); rem1 = n1 - 256 * (K1 - 1);           // entries of the last step of each half (1..256; unused when K == 0)
    }
    __host__ __device__ int position(int s) const {
        const int c = s >= n0;
        const int sc = c ? s - n0 : s, K = c ? K1 : K0, rem = c ? rem1 : rem0;
        int t, j;
        if (sc < rem * K) { t = sc / K; j = sc - t * K; }
        else { const int q = sc - rem * K; t = rem + q / (K - 1); j = q - (t - rem) * (K - 1); }
        return (c ? n0 : 0) + j * 256 + t;
    }
};
constexpr int SORT_NT = 256, SORT_NW = SORT_NT / 64, SORT_KEYS = TS * TS, SORT_KEY_BITS = 10;
static_assert(SORT_KEYS == 1 << SORT_KEY_BITS, "pixkey has SORT_KEY_BITS bits");
__device__ __forceinline__ uint32_t pixkey(uint32_t xy) { return ((xy >> 11) & (31u << 5)) | (xy & 31u); }     // pixel inside the 32x32 tile (raster order)
__global__ __launch_bounds__(SORT_NT) void k_segsort(int n_items, const Item* __restrict__ items,
                                                      const uint32_t* __restrict__ src_xy, const double* __restrict__ src_t,
                                                      uint32_t* __restrict__ dst_xy, double* __restrict__ dst_t)
{
    __shared__ uint32_t cnt[SORT_NW][SORT_KEYS];        // per (wave, key): count, then the running output rank
    __shared__ uint32_t wsum[SORT_NT];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int k = blockIdx.x; k < n_items; k += gridDim.x) {
        const Item it = items[k];
        const int n = it.count, begin = it.begin;
        const int qlen = (((n + SORT_NW - 1) / SORT_NW + 63) / 64) * 64;           // a wave's share of the input: whole wave steps
        const int lo = min(wv * qlen, n), hi = min(lo + qlen, n);
        for (int i = threadIdx.x; i < SORT_NW * SORT_KEYS; i += SORT_NT) (&cnt[0][0])[i] = 0u;
        __syncthreads();
        for (int i = lo + lane; i < hi; i += 64) atomicAdd(&cnt[wv][pixkey(src_xy[begin + i])], 1u);
        __syncthreads();
        // exclusive scan over (key, wave), key-major: thread t owns keys 4t .. 4t+3
        uint32_t c[4][SORT_NW], tot = 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < SORT_NW; ++w) { c[q][w] = cnt[w][4 * threadIdx.x + q]; tot += c[q][w]; }
        wsum[threadIdx.x] = tot;
        __syncthreads();
        for (int off = 1; off < SORT_NT; off <<= 1) {          // Hillis-Steele inclusive scan of the 256 thread totals
            const uint32_t v = (threadIdx.x >= off) ? wsum[threadIdx.x - off] : 0u;
            __syncthreads();
            wsum[threadIdx.x] += v;
            __syncthreads();
        }
        uint32_t run = wsum[threadIdx.x] - tot;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int w = 0; w < SORT_NW; ++w) { cnt[w][4 * threadIdx.x + q] = run; run += c[q][w]; }
        __syncthreads();
        // placement: a wave walks its quarter in input order; lanes of a step that share a key are ranked by lane
        const SegLayout L(n);
        // (the next step's events are fetched before this step is ranked: the steps of a wave depend on each other through LDS only)
        uint32_t nxy = 0u; double nt = 0.0;
        if (lo + lane < hi) { nxy = src_xy[begin + lo + lane]; nt = src_t[begin + lo + lane]; }
        for (int i0 = lo; i0 < hi; i0 += 64) {                  // wave-uniform trip count
            const int i = i0 + lane;
            const bool valid = i < hi;
            const uint32_t xy = nxy; const double t = nt;
            if (i + 64 < hi) { nxy = src_xy[begin + i + 64]; nt = src_t[begin + i + 64]; }
            const int key = valid ? (int)pixkey(xy) : -1;
            // the lanes that share a key, found bit by bit (SORT_KEYS = 2^10: ten ballots instead of one LDS round per distinct key of the step)
            unsigned long long grp = __ballot(valid);
#pragma unroll
            for (int bit = 0; bit < SORT_KEY_BITS; ++bit) {
                const bool one = ((key >> bit) & 1) != 0;
                const unsigned long long bal = __ballot(one);
                grp &= one ? bal : ~bal;
            }
            uint32_t rank = 0u, base = 0u;
            if (valid) { base = cnt[wv][key]; rank = base + (uint32_t)__popcll(grp & lt_mask); }      // every lane reads before a leader updates
            __builtin_amdgcn_wave_barrier();
            if (valid && (grp & lt_mask) == 0ull) cnt[wv][key] = base + (uint32_t)__popcll(grp);
            __builtin_amdgcn_wave_barrier();
            if (valid) {
                const int pos = L.position((int)rank);
                dst_xy[begin + pos] = xy;
                dst_t[begin + pos] = t;
            }
        }
        __syncthreads();                                        // cnt is reused by the next segment of this workgroup
    }
}

// thread per (window, tile): emit its segments
__global__ void k_items(Geom g, int seg, const int32_t* __restrict__ tilecount, const int32_t* __restrict__ tilebase,
                        const int32_t* __restrict__ itembase, Item* __restrict__ items)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= g.B * g.ntiles) return;
    const int c = tilecount[idx], base = tilebase[idx];
    int k = itembase[idx];
    const int len = balanced_seg_len(c, seg);            // same number of segments as ceil(c/seg) (k_bin_tilescan), equal lengths
    for (int s = 0; s < c; s += len, ++k) {
        Item it;
        it.win = idx / g.ntiles; it.tile = idx % g.ntiles; it.begin = base + s; it.count = min(len, c - s);
        it.t_lo = 0.0; it.t_hi = 0.0;
        items[k] = it;
    }
}

// workgroup per segment (grid-stride): exact time range of its events
__global__ __launch_bounds__(NT) void k_seg_minmax(int n_items, Item* __restrict__ items, const double* __restrict__ ev_t)
{
    __shared__ double smn[NWAVE], smx[NWAVE];
    for (int k = blockIdx.x; k < n_items; k += gridDim.x) {
        const int begin = items[k].begin, count = items[k].count;
        double mn = INFINITY, mx = -INFINITY;
        for (int i = threadIdx.x; i < count; i += NT) { const double t = ev_t[begin + i]; mn = fmin(mn, t); mx = fmax(mx, t); }
        mn = wave_min(mn); mx = wave_max(mx);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < NWAVE; ++i) { mn = fmin(mn, smn[i]); mx = fmax(mx, smx[i]); }
            items[k].t_lo = mn; items[k].t_hi = mx;
        }
    }
}

// edges (B,R,H,W) double -> float, and the fp64 moments of the STORED values: sum E, sum E^2 (and max |E|) per (b, r), as one partial triple
// per block (the host adds the EDGE_PARTS pairs in index order: no atomics, bit-reproducible).  grid (EDGE_PARTS, R, B).
constexpr int EDGE_PARTS = 32;
constexpr int EDGE_MOM = 3;               // per part: sum E, sum E^2, max |E|
__global__ __launch_bounds__(NT) void k_edges(Geom g, const double* __restrict__ src, float* __restrict__ dst, double* __restrict__ moments)
{
    __shared__ double scratch[NWAVE];
    const int r = blockIdx.y, b = blockIdx.z;
    const size_t n = (size_t)g.H * g.W, base = ((size_t)b * g.R + r) * n;
    double s = 0.0, ss = 0.0, mx = 0.0;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const float f = (float)src[base + i];
        dst[base + i] = f;
        s += (double)f; ss += (double)f * (double)f;
        mx = fmax(mx, fabs((double)f));             // fmax drops a NaN: a NaN edge map surfaces in the sums
    }
    s = block_sum(s, scratch);
    ss = block_sum(ss, scratch);
    mx = wave_max(mx);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NWAVE; ++i) mx = fmax(mx, scratch[i]);
        double* o = moments + (((size_t)b * g.R + r) * EDGE_PARTS + blockIdx.x) * EDGE_MOM;
        o[0] = s; o[1] = ss; o[2] = mx;
    }
}

// Per window: first segment index (segments are ordered by window) and dtmax = max |t - tau_r| over its events and reference
// times, from the segments' exact time ranges.  grid (B), one workgroup each.  itembase: (B*ntiles) first segment of each tile.
__global__ __launch_bounds__(NT) void k_win_consts(Geom g, int n_items, const Item* __restrict__ items, const int32_t* __restrict__ itembase,
                                                    const double* __restrict__ edge_ts, int32_t* __restrict__ win_item0,
                                                    double* __restrict__ dtmax)
{
    __shared__ double smn[NWAVE], smx[NWAVE];
    const int b = blockIdx.x;
    const int lo = itembase[(size_t)b * g.ntiles], hi = (b + 1 < g.B) ? itembase[(size_t)(b + 1) * g.ntiles] : n_items;
    double mn = INFINITY, mx = -INFINITY;
    for (int k = lo + threadIdx.x; k < hi; k += NT) { mn = fmin(mn, items[k].t_lo); mx = fmax(mx, items[k].t_hi); }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) { smn[threadIdx.x >> 6] = mn; smx[threadIdx.x >> 6] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < NWAVE; ++i) { mn = fmin(mn, smn[i]); mx = fmax(mx, smx[i]); }
        double d = 0.0;
        if (hi > lo)
            for (int r = 0; r < g.R; ++r) { const double tau = edge_ts[b * g.R + r]; d = fmax(d, fmax(fabs(mn - tau), fabs(mx - tau))); }
        win_item0[b] = lo;
        dtmax[b] = d;
    }
}

}  // namespace eincm
