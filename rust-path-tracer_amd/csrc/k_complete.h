/*
 * k_complete.h — the completion of sample generations (kernels/src/lib.rs:225-226: `output[i] += (radiance, 1)`, `rng[i].x += 1`), the one kernel that adds
 * finished samples to the accumulators.  Compiled in rpt_hip.hip.
 */
#ifndef RPT_K_COMPLETE_H
#define RPT_K_COMPLETE_H

#include "k_path.h"

/* Completion of generations: ONE WAVE PER CHUNK of 64 pixels, lane = pixel.  When all S slots of a pixel are finished (HIT_DONE) or have
 * nothing left (HIT_IDLE) and at least one is finished, their radiances are added to the accumulator IN SLOT ORDER (= sample order,
 * kernels/src/lib.rs:225, src/trace.rs:295: the f32 sum order is part of the result), the pixel's rng.n advances by the number of samples
 * (lib.rs:226), and every finished slot starts its next sample (lib.rs:36-60) or goes idle.
 *
 * Every global access is a ROW of the chunk — 64 consecutive slots, one coalesced wave access — whatever the slot layout (k_common.h
 * slot_pix: a row is 64 pixels at one sample index for q_shift = 0, 64 / Q pixels x Q samples otherwise):
 *   pass 1  the hit words, row by row: two ballots per row tell every lane which of ITS pixel's slots are finished / still in flight;
 *   pass 2  q_shift = 0: a row is one sample of the chunk's 64 pixels, lane = pixel — eight rows in flight, added in order from the registers;
 *           q_shift > 0: the radiance records of up to 32 samples of all 64 pixels, row by row into an LDS tile [sample][pixel] (the transposition
 *           the layout needs), then every lane adds its pixel's column of the tile in sample order;
 *           either way the finished slots of a row go idle (or start their next sample) as the row passes.
 * Slot k takes the samples k, k + S, ... of a call, so within a generation the finished slots of a pixel are a PREFIX of its slots: the lane
 * keeps a count (checked against the highest finished slot), which is what lets S go beyond the 32 bits of a mask — up to 256 samples of a
 * pixel in flight (a rank that owns 1/8 of an image fills its launches with 8 x the samples per pixel).
 * History: round 2 completed inside the shade stage (a DPP chain across lanes, 0.45 ms of a DarkCornell batch); rounds 3-5 one THREAD per
 * pixel looping over its slots — coalesced only at q_shift = 0: 5.6 ms instead of 0.9 per 2048^2 x 32 slots on the large-scene layout.
 * A batch of known length (no slot takes a second sample) runs this once, after its last iteration; otherwise it follows every shade stage. */
#define RPT_COMPLETE_ROWS 32u          /* samples of a pixel staged per pass */
#define RPT_COMPLETE_PITCH 65u         /* float4 per tile row: 64 pixels + 1 (the transposing stores of a q_shift > 0 row would share a bank) */
__host__ __device__ __forceinline__ uint32_t complete_rows(uint32_t S) { return S < RPT_COMPLETE_ROWS ? S : RPT_COMPLETE_ROWS; }
__host__ __device__ __forceinline__ size_t complete_lds_bytes(uint32_t S, uint32_t q_shift) {
    return q_shift == 0u ? 0u : (size_t)complete_rows(S) * RPT_COMPLETE_PITCH * sizeof(float4) + (size_t)S * sizeof(unsigned long long);
}
/* pass 1 of k_complete for G consecutive rows of the chunk (G loads in flight): the row's "finished" ballot goes to LDS for pass 2, the lane
 * takes the bits of its own pixel out of the rows that hold it */
template <uint32_t G>
__device__ __forceinline__ void complete_status_rows(const DevState &st, uint32_t base, uint32_t j0, uint32_t lane, unsigned long long *row_done, uint32_t my_group,
                                                     uint32_t my_shift, unsigned long long q_mask, uint32_t &n_done, uint32_t &n_busy, uint32_t &top) {
    const uint32_t gs = st.group_shift, qs = st.q_shift;
    uint32_t w[G];
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) w[i] = __float_as_uint(st.hit[base + ((j0 + i) << 6) + lane].y);
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) {
        const uint32_t row = j0 + i;
        const bool done = w[i] == HIT_DONE;
        const unsigned long long md = rpt_ballot(done), mb = rpt_ballot(!done && w[i] != HIT_IDLE);
        if (qs != 0u && lane == 0u) row_done[row] = md;               /* (q_shift = 0: pass 2 needs no masks, the finished slots of a pixel are its first n_done) */
        if ((row >> (gs - qs)) == my_group) {
            const uint32_t bits = (uint32_t)((md >> my_shift) & q_mask);
            n_done += (uint32_t)__popc(bits);
            n_busy += (uint32_t)__popc((uint32_t)((mb >> my_shift) & q_mask));
            if (bits != 0u) top = ((row & ((1u << (gs - qs)) - 1u)) << qs) + 32u - (uint32_t)__clz((int)bits);    /* (rows ascend: the last one wins) */
        }
    }
}
/* pass 2 at q_shift = 0 (every shipped scene): row k of the chunk IS sample k of its 64 pixels with the lane's own pixel in its own lane — nothing to
 * transpose, no tile: G rows in flight, added in order straight from the registers; finished slots (k < n_done: the prefix) that owe nothing go idle. */
template <uint32_t G>
__device__ __forceinline__ bool complete_direct_rows(const DevState &st, uint32_t base, uint32_t k0, uint32_t lane, bool ok, uint32_t n_done, float4 &acc) {
    float rx[G], ry[G], rz[G], rw[G];
    bool restart = false;
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) {
        const float4 r = st.rad[base + ((k0 + i) << 6) + lane];
        rx[i] = r.x; ry[i] = r.y; rz[i] = r.z; rw[i] = r.w;
    }
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) {
        const bool mine = ok && k0 + i < n_done;
        if (mine) { acc.x += rx[i]; acc.y += ry[i]; acc.z += rz[i]; acc.w += 1.0f; }
        const uint32_t todo = __float_as_uint(rw[i]);
        if (mine && todo == 0u) st.hit[base + ((k0 + i) << 6) + lane] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        restart = restart || rpt_ballot(mine && todo != 0u) != 0ull;
    }
    return restart;
}
/* Row t of a block of `rows` samples starting at sample kb: for every pixel group, rows >> qs consecutive rows of the chunk. */
__device__ __forceinline__ uint32_t complete_block_slot(const DevState &st, uint32_t base, uint32_t kb, uint32_t t, uint32_t rows, uint32_t rows_log, uint32_t lane) {
    const uint32_t gs = st.group_shift, qs = st.q_shift;
    const uint32_t g = t >> (rows_log - qs), jl = (kb >> qs) + (t & ((rows >> qs) - 1u));
    return base + (((g << (gs - qs)) | jl) << 6) + lane;
}
/* pass 2 of k_complete for G rows of such a block (G loads in flight): the radiance records into the tile [sample - kb][pixel], the finished slots of the
 * completing pixels that owe nothing more marked idle.  Returns (wave-uniform) whether some finished slot owes another sample (complete_restart_rows). */
template <uint32_t G>
__device__ __forceinline__ bool complete_stage_rows(const DevState &st, uint32_t base, uint32_t kb, uint32_t t0, uint32_t rows, uint32_t rows_log, uint32_t lane, float4 *tile,
                                                    const unsigned long long *row_done, unsigned long long ok_mask) {
    float rx[G], ry[G], rz[G], rw[G];              /* (scalars: an array of float4 stays in scratch behind its 16-byte copies) */
    uint32_t slot_of[G];
    bool restart = false;
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) {
        slot_of[i] = complete_block_slot(st, base, kb, t0 + i, rows, rows_log, lane);
        const float4 r = st.rad[slot_of[i]];
        rx[i] = r.x; ry[i] = r.y; rz[i] = r.z; rw[i] = r.w;
    }
#pragma unroll
    for (uint32_t i = 0u; i < G; ++i) {
        const uint32_t slot = slot_of[i], p = slot_pix(st, slot) & 63u, k = slot_k(st, slot);
        tile[(k - kb) * RPT_COMPLETE_PITCH + p] = make_float4(rx[i], ry[i], rz[i], rw[i]);
        const bool mine = ((row_done[(slot - base) >> 6] >> lane) & 1ull) != 0ull && ((ok_mask >> p) & 1ull) != 0ull;
        const uint32_t todo = __float_as_uint(rw[i]);
        if (mine && todo == 0u) st.hit[slot] = make_float2(0.0f, __uint_as_float(HIT_IDLE));
        restart = restart || rpt_ballot(mine && todo != 0u) != 0ull;
    }
    return restart;
}
/* the finished slots of a block that owe another sample start it (lib.rs:36-60) — never in the one completion of a batch of known length.  From the tile
 * (the record's .w is what the slot owes), row by row; slot k takes the samples k, k + S, ... */
__device__ __forceinline__ bool complete_restart_rows(const DevState &st, const DevConfig &cfg, uint32_t base, uint32_t kb, uint32_t rows, uint32_t rows_log, uint32_t lane,
                                                      const float4 *tile, const unsigned long long *row_done, unsigned long long ok_mask, uint32_t new_n, uint32_t rng_offset) {
    bool started = false;
#pragma unroll 1
    for (uint32_t t = 0u; t < rows; ++t) {
        const uint32_t slot = complete_block_slot(st, base, kb, t, rows, rows_log, lane), p = slot_pix(st, slot) & 63u, k = slot_k(st, slot);
        const uint32_t todo = __float_as_uint(tile[(k - kb) * RPT_COMPLETE_PITCH + p].w);
        const bool mine = ((row_done[(slot - base) >> 6] >> lane) & 1ull) != 0ull && ((ok_mask >> p) & 1ull) != 0ull && todo != 0u;
        const uint32_t n_of = (uint32_t)__shfl((int)new_n, (int)p, RPT_WAVE), offset_of = (uint32_t)__shfl((int)rng_offset, (int)p, RPT_WAVE);
        if (mine) {
            start_path(st, cfg, slot, n_of + k, offset_of, todo - 1u);
            started = true;
        }
    }
    return started;
}
__global__ __launch_bounds__(RPT_WAVE) void k_complete(DevState st, DevQueues q, DevConfig cfg, uint32_t iteration, uint32_t final_pass,
                                                       DevStats *stats) {
    /* a surplus launch of the run-ahead returns at once (grid-uniform) — but not the one completion of a batch of known length:
     * "drained" there only says that no RAY was left in an earlier iteration, the finished samples still wait to be added */
    if (!final_pass && q.count[Q_DRAINED] != 0u) return;
    extern __shared__ float4 complete_lds[];
    const uint32_t gs = st.group_shift, qs = st.q_shift, S = 1u << gs, rows = complete_rows(S);
    float4 *tile = complete_lds;
    unsigned long long *row_done = reinterpret_cast<unsigned long long *>(complete_lds + rows * RPT_COMPLETE_PITCH);
    const uint32_t lane = threadIdx.x, base = blockIdx.x << (6u + gs), pix = (blockIdx.x << 6) | lane;
    const bool in_image = pix < st.n_pixels;
    /* the rows that hold this lane's pixel (row >> (gs - qs) == its group), and where its Q slots sit in such a row's ballots */
    const uint32_t my_group = lane >> (6u - qs), my_shift = (lane & ((64u >> qs) - 1u)) << qs;
    const unsigned long long q_mask = (1ull << (1u << qs)) - 1ull;             /* (Q <= 32) */
    uint32_t n_done = 0u, n_busy = 0u, top = 0u;
    bool nothing_to_do = false;
    if (S >= 8u) {
        for (uint32_t j0 = 0u; j0 < S && !nothing_to_do; j0 += 8u) {
            complete_status_rows<8>(st, base, j0, lane, row_done, my_group, my_shift, q_mask, n_done, n_busy, top);
            /* between the iterations of a call whose slots take several samples most pixels have a sample in flight: nothing to do for the chunk */
            nothing_to_do = !final_pass && rpt_ballot(in_image && n_busy == 0u) == 0ull;
        }
    } else if (S == 4u) complete_status_rows<4>(st, base, 0u, lane, row_done, my_group, my_shift, q_mask, n_done, n_busy, top);
    else complete_status_rows<2>(st, base, 0u, lane, row_done, my_group, my_shift, q_mask, n_done, n_busy, top);
    if (nothing_to_do) return;
    const bool ok = in_image && n_busy == 0u && n_done != 0u && top == n_done;
    if (final_pass && in_image && (n_busy != 0u || top != n_done)) {
        /* the one completion of a batch of known length found a sample still in flight: the bound on its iterations was wrong (must never
         * happen; rpt_wait / rpt_render report it) — or finished slots that are no prefix.  Counted like k_check_drained would: slots not idle. */
        atomicAdd(&stats->undrained, (unsigned long long)(n_done + n_busy));
    }
    const unsigned long long ok_mask = rpt_ballot(ok);
    bool started = false;
    if (ok_mask != 0ull) {
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        uint2 rs = make_uint2(0u, 0u);
        if (ok) { acc = st.accum[pix]; rs = st.rng[pix]; }
        const uint32_t new_n = rs.x + n_done;
        uint32_t most = ok ? n_done : 0u;                                   /* the chunk's longest prefix (wave-uniform) */
        for (uint32_t o = 32u; o != 0u; o >>= 1) { const uint32_t other = (uint32_t)__shfl_xor((int)most, (int)o, RPT_WAVE); most = other > most ? other : most; }
        const uint32_t rows_log = 31u - (uint32_t)__clz((int)rows);
        __syncthreads();                                                    /* row_done written */
        if (qs == 0u) {
            const uint32_t G = S >= 8u ? 8u : S;
            for (uint32_t k0 = 0u; k0 < most; k0 += G) {
                bool restart;
                if (G == 8u) restart = complete_direct_rows<8>(st, base, k0, lane, ok, n_done, acc);
                else if (G == 4u) restart = complete_direct_rows<4>(st, base, k0, lane, ok, n_done, acc);
                else restart = complete_direct_rows<2>(st, base, k0, lane, ok, n_done, acc);
                if (restart) {                                              /* (never in the one completion of a batch of known length) */
#pragma unroll 1
                    for (uint32_t k = k0; k < k0 + G; ++k) {
                        const uint32_t slot = base + (k << 6) + lane, todo = __float_as_uint(st.rad[slot].w);
                        if (ok && k < n_done && todo != 0u) {
                            start_path(st, cfg, slot, new_n + k, rs.y, todo - 1u);      /* slot k takes the samples k, k + S, ... */
                            started = true;
                        }
                    }
                }
            }
        } else
        for (uint32_t kb = 0u; kb < most; kb += rows) {
            /* the rows that hold samples [kb, kb + rows) of all 64 pixels: for every pixel group, rows >> qs consecutive rows */
            bool restart = false;
            if (rows >= 8u) {
                for (uint32_t t0 = 0u; t0 < rows; t0 += 8u) restart |= complete_stage_rows<8>(st, base, kb, t0, rows, rows_log, lane, tile, row_done, ok_mask);
            } else if (rows == 4u) restart = complete_stage_rows<4>(st, base, kb, 0u, rows, rows_log, lane, tile, row_done, ok_mask);
            else restart = complete_stage_rows<2>(st, base, kb, 0u, rows, rows_log, lane, tile, row_done, ok_mask);
            __syncthreads();
            const uint32_t here = most - kb < rows ? most - kb : rows;
            for (uint32_t kl = 0u; kl < here; ++kl) {
                const float4 r = tile[kl * RPT_COMPLETE_PITCH + lane];
                if (ok && kb + kl < n_done) { acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += 1.0f; }
            }
            if (restart) started |= complete_restart_rows(st, cfg, base, kb, rows, rows_log, lane, tile, row_done, ok_mask, new_n, rs.y);
            __syncthreads();
        }
        if (ok) {
            st.accum[pix] = acc;
            rs.x = new_n;
            st.rng[pix] = rs;
        }
    }
    /* tell the host that new samples were started (one plain store per wave, every writer stores 1) */
    const unsigned long long any = rpt_ballot(started);
    if (any != 0ull && lane == (uint32_t)__ffsll((long long)any) - 1u) raise_flag(&q.count[Q_REGEN0 + (iteration & 1u) * Q_LINE]);
}

#endif /* RPT_K_COMPLETE_H */
