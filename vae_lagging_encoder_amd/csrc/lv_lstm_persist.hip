// lv_lstm_persist.hip -- the LSTM recurrences as ONE persistent launch each (bf16 recurrent operands, H = 1024).
//
// lv_lstm.hip pays one kernel boundary per timestep and re-reads its slice of W_hh from L2 / Infinity Cache every step
// (measured: ~4.5 us per forward step, 7.7 us per BPTT step).  Here the chip is cut along its XCDs instead:
//
//   * 256 workgroups, one per CU, in 8 groups of 32 (group = blockIdx % 8: one XCD under the usual round-robin
//     placement -- only speed depends on that, never correctness).  A group owns a slice of the BATCH (rows
//     [g*R, g*R+R), R = ceil(B/8)) and carries it through all T steps on its own; groups never talk.
//   * every wave keeps its 64 KB slice of W_hh in REGISTERS for the whole call (256 VGPRs per lane; one wave per SIMD), so
//     after the prologue no weight byte moves again.
//   * per step the only traffic is a hand-off inside the group, as 8-byte tagged granules written with agent-scope relaxed
//     64-bit stores (write-through) and read by polling the tags: no fences, no flags -- the tag travels with the data.
//
// Four kernels (the K-split forward and the reduce-scatter BPTT are the defaults; the other two are kept for A/B and tests):
//   lstm_fwd_persist_kernel     forward, gate COLUMNS split over the waves: all-gather of h_t ({2 x bf16, tag} granules, 2048
//                               per workgroup and step at R = 4) into an LDS image, barrier, 64 x v_mfma_f32_16x16x32_bf16.
//   lstm_fwd_persist_ks_kernel  forward, CONTRACTION split over the waves: each wave gathers only its K quarter and multiplies
//                               it at once (128 x v_mfma_f32_4x4x4_16b_bf16), the quarter products meet after one barrier.
//   lstm_bwd_persist_kernel     BPTT, all-gather of dG[t] (4 gate gradients per unit: 8192 granules per workgroup and step).
//   lstm_bwd_persist_rs_kernel  BPTT, reduce-scatter: a workgroup multiplies ITS units' dG with its 128 gate rows of W_hh and
//                               sends every workgroup the partial sums of the 32 units it owns ({2 x 28-bit float, tag}
//                               granules): 2048 granules received per workgroup and step, contraction on the 4x4x4 MFMA.
//   Measured per timestep at B = 32 (R = 4): 3.15 / 3.06 / 5.32 / 3.0 us.
//
// Every spin is bounded: a workgroup that waits longer than ~1 s raises *status and the whole launch drains.
// Requirements (else LV_ERR_UNSUPPORTED and the caller uses the launch-per-step kernels): H == 1024, B <= 64 (forward) /
// 32 (BPTT), a 256-CU device (all 256 workgroups must be resident at once), gx in unit-major column order.
// The packed weight images are built by lv_lstm_persist_pack and passed in: a caller whose weights do not change between
// calls (the decoder during the aggressive inner loop) packs once.  The same sources run on the CPU emulator with every
// workgroup live as fibers (tests/emu): LV_LAUNCH_RESIDENT / LV_BLOCK_SHARED / lv_agent_* in lv_device.h.
#include "lv_device.h"
#include "lv_persist_common.h"

namespace {

using namespace lvp;

constexpr int PKS = PH / 32;        // MFMA k-steps per product
constexpr int PUW = 8;              // hidden units per wave
constexpr int HPITCH = PH / 2 + 16; // LDS row pitch of the gathered h image in dwords: rows 16 banks apart, so the A-fragment
                                    // reads of 4 rows x 4 k-quads hit 16 distinct bank groups
constexpr int PRMAX = 8;            // batch rows per group this build supports (LDS: 2 x PRMAX x HPITCH dwords)
// forward granules: (tag << 32) | two bf16

// Wpk[wave_id (128)][ks (32)][nb (2)][lane (64)] : lane (c = l&15, kq = l>>4) holds W_hh[gate*H + unit][32ks + 8kq .. +7]
// with unit = 8*wave_id + 4*nb + (c>>2), gate = c&3 -- the B operand of v_mfma_f32_16x16x32_bf16 for that column.
__global__ __launch_bounds__(256) void pack_w_persist_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 128L * PKS * 2 * 64) return;
    const int l = (int)(idx & 63);
    const int nb = (int)((idx >> 6) & 1);
    const int ks = (int)((idx >> 7) % PKS);
    const int wave_id = (int)(idx / (64L * 2 * PKS));
    const int c = l & 15, kq = l >> 4;
    const int unit = PUW * wave_id + 4 * nb + (c >> 2), gate = c & 3;
    const float* row = whh + ((long)gate * PH + unit) * PH + 32 * ks + 8 * kq;
    wpk[idx] = make_uint4(lv_pack_bf16x2(row[0], row[1]), lv_pack_bf16x2(row[2], row[3]), lv_pack_bf16x2(row[4], row[5]),
                          lv_pack_bf16x2(row[6], row[7]));
}

struct PersistFwdP {
    const float* gx;            // [T][B][H][4] unit-major gate pre-activations (input projection + biases)
    const uint4* wpk;
    float* hs; float* cs;       // [T+1][B][H], index 0 = initial state
    float* gates;               // [T][B][H][4] records for BPTT
    const uint8_t* dmask; float dscale; float* hdrop;
    gran_t* hx;                 // exchange: [2 parity][8 groups][16 rows][H/2] granules, zeroed before the launch
    int* status;
    int T, B, R;
};

constexpr int SB = 8;               // timesteps per I/O block (see below)

// Global loads and stores of a wave retire in order on gfx9 (one vmcnt), so ANY load or store issued inside a step
// ends up in front of the next hand-off poll and the poll waits for it (measured: +2.3 us per step for the result
// stores alone).  The recurrence therefore does its bulk I/O in blocks of SB steps: the gate pre-activations of the
// next block are fetched and the results of the previous block are written at block boundaries, and in between a
// step touches global memory for the hand-off only.  Each lane owns ONE (batch row, unit) pair for the whole call.
struct __attribute__((aligned(16))) FwdLds {
    uint32_t hl[2][PRMAX * HPITCH];     // gathered h_{t-1}, [parity][row][k/2]
    float pre[4][16][33];               // per wave: MFMA tile [row][col]
    int abort;
};
__global__ __launch_bounds__(256) void lstm_fwd_persist_kernel(PersistFwdP p) {
    LV_BLOCK_SHARED(FwdLds, sm);
    uint32_t (&hl)[2][PRMAX * HPITCH] = sm.hl;
    float (&pre)[4][16][33] = sm.pre;
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);          // valid batch rows of this group
    if (rows == 0) return;                                                    // (uniform per group: nobody waits on it)
    if (tid == 0) s_abort = 0;

    // ---- weights: registers for the whole call --------------------------------------------------------------------------
    uint4 wreg[PKS][2];
    {
        const uint4* wp = p.wpk + (long)wave_id * PKS * 2 * 64 + l;
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wreg[ks][nb] = wp[(ks * 2 + nb) * 64];
    }

    // ---- this lane's (row, unit) pair -------------------------------------------------------------------------------------
    const int prow = l >> 3, ul = l & 7;
    const int punit = PUW * wave_id + ul;
    const bool own = prow < rows;
    const long BH = (long)B * PH;
    const long pidx = (long)(b0 + (own ? prow : 0)) * PH + punit;          // index into a [B][H] slab
    float c_state = own ? p.cs[pidx] : 0.f;
    gran_t* const hx_g = p.hx + (long)group * 16 * (PH / 2);
    const long hx_par = (long)PGROUPS * 16 * (PH / 2);
    gran_t* const my_gran = hx_g + (long)prow * (PH / 2) + (punit >> 1);
    const bool publisher = own && !(ul & 1);

    {   // publish the initial state hs[0] as state 0 (tag 1)
        const uint32_t mine = lv_f32_to_bf16_bits(own ? p.hs[pidx] : 0.f);
        const uint32_t next = (uint32_t)__shfl_down((int)mine, 1, 64);
        if (publisher) gran_store(my_gran, ((gran_t)1u << 32) | (gran_t)(mine | (next << 16)));
    }

    float4 gxb[SB];                  // gate pre-activations of the current block's steps
    float keepb[SB];
    float4 recb[SB];                 // results of the current block: gate record, c, h, dropped h
    float cb[SB], hb[SB], hdb[SB];
    auto load_block = [&](int tb) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            gxb[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            keepb[s2] = 1.f;
            if (own && t < T) {
                gxb[s2] = *reinterpret_cast<const float4*>(p.gx + ((long)t * BH + pidx) * 4);
                if (p.hdrop && p.dmask) keepb[s2] = p.dmask[((long)(b0 + prow) * T + t) * PH + punit] ? p.dscale : 0.f;
            }
        }
    };
    auto store_block = [&](int tb) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            if (own && t < T) {
                // streaming stores: the results are consumed by later kernels only, and a write-allocating store of a
                // partial line makes L2 fetch the line first -- measured 3.85 -> 3.46 us per step with the nt hint
                lv_store_nt(f32x4{recb[s2].x, recb[s2].y, recb[s2].z, recb[s2].w},
                            reinterpret_cast<f32x4*>(p.gates + ((long)t * BH + pidx) * 4));
                lv_store_nt(cb[s2], p.cs + (long)(t + 1) * BH + pidx);
                lv_store_nt(hb[s2], p.hs + (long)(t + 1) * BH + pidx);
                if (p.hdrop) lv_store_nt(hdb[s2], p.hdrop + (long)t * BH + pidx);
            }
        }
    };
    load_block(0);
    __syncthreads();

    // A-operand rows beyond the group's batch rows read a valid LDS row (row 0): their products land in MFMA output rows
    // nobody owns, so no predicate sits between the LDS reads and the MFMAs
    const int arow = (l & 15) < rows ? (l & 15) : 0, kq = l >> 4;
    const int ngran = rows * (PH / 2);                 // granules of one state of this group
    const int gq = (ngran + 3) / 4;                    // this wave gathers granules [w*gq, w*gq + gq)

    for (int tb = 0; tb < T; tb += SB) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            if (t >= T) break;
            // ---- gather state t (tag t+1) of the whole group into LDS ---------------------------------------------------
            const gran_t* src = hx_g + (long)(t & 1) * hx_par;
            uint32_t* dst = hl[t & 1];
            const uint32_t want = (uint32_t)(t + 1);
            for (int base = w * gq; base < w * gq + gq; base += 64 * 8) {
                gran_t v[8];
                int spins = 0;
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int idx = base + j * 64 + l;
                        const bool in = idx < w * gq + gq && idx < ngran;
                        v[j] = gran_load(src + (in ? idx : 0));          // out-of-range lanes re-read granule 0: no branch
                        ok = ok && (!in || (uint32_t)(v[j] >> 32) == want);
                    }
                    ok = __all(ok);
                    if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; break; }
                } while (!ok);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int idx = base + j * 64 + l;
                    if (idx < w * gq + gq && idx < ngran) dst[(idx >> 9) * HPITCH + (idx & 511)] = (uint32_t)v[j];
                }
            }
            __syncthreads();
            if (s_abort) { if (tid == 0) atomicExch(p.status, 100 + t); return; }

            // ---- recurrent product: this wave's 32 gate columns, K from LDS x registers ---------------------------------
            f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            const uint4* arowp = reinterpret_cast<const uint4*>(dst + arow * HPITCH) + kq;
#pragma unroll
            for (int ks = 0; ks < PKS; ++ks) {
                const uint4 a = arowp[ks * 4];
                acc[(ks & 1) * 2 + 0] = lv_mfma_16x16x32_bf16(a, wreg[ks][0], acc[(ks & 1) * 2 + 0]);
                acc[(ks & 1) * 2 + 1] = lv_mfma_16x16x32_bf16(a, wreg[ks][1], acc[(ks & 1) * 2 + 1]);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[w][(l >> 4) * 4 + r][nb * 16 + (l & 15)] = acc[nb][r] + acc[2 + nb][r];
            LV_WAIT_LDS();        // lgkmcnt(0): wave-private tile, the wave's own LDS accesses are ordered

            // ---- epilogue: gates, cell update, hand-off of h_t ------------------------------------------------------------
            float h = 0.f;
            if (own) {
                const float* pr = &pre[w][prow][4 * ul];
                const float ig = lv_sigmoid_fast(gxb[s2].x + pr[0]), fg = lv_sigmoid_fast(gxb[s2].y + pr[1]);
                const float gg = lv_tanh_fast(gxb[s2].z + pr[2]), og = lv_sigmoid_fast(gxb[s2].w + pr[3]);
                const float c = fg * c_state + ig * gg;
                h = og * lv_tanh_fast(c);
                c_state = c;
                recb[s2] = make_float4(ig, fg, gg, og);
                cb[s2] = c; hb[s2] = h; hdb[s2] = h * keepb[s2];
            }
            const uint32_t mine = lv_f32_to_bf16_bits(h);
            const uint32_t next = (uint32_t)__shfl_down((int)mine, 1, 64);
            if (publisher)
                gran_store(my_gran + (long)((t + 1) & 1) * hx_par, ((gran_t)(uint32_t)(t + 2) << 32) | (gran_t)(mine | (next << 16)));
        }
        // ---- block boundary: the only bulk global traffic of the recurrence ---------------------------------------------
        store_block(tb);
        load_block(tb + SB);
    }
}

// =====================================================================================================================
// Forward recurrence, K-split form.  Same groups, same hand-off (all-gather of h_t as {2 x bf16, tag} granules); what changes
// is who multiplies what.  A workgroup owns 32 units = 128 gate columns and its 4 waves split the CONTRACTION: wave w gathers
// only K-quarter w of h_{t-1} (units [256w, 256w+256): R x 128 granules, into a wave-private part of the LDS image), starts
// multiplying as soon as ITS quarter has landed -- no workgroup barrier between gather and product -- and contracts on
// v_mfma_f32_4x4x4_16b_bf16 (the batch slice is 4 rows: half the matrix-pipe time of the 16-row tile; 128 steps per wave, 32
// LDS reads of the lane's A row in hand-pipelined batches).  The four quarter products [R][128] meet in LDS, ONE barrier,
// and each (row, unit) lane adds its 4 x 4 values and runs the gate math.  The barrier sits after the product instead of
// before it, so the skew between the four gathers is absorbed by the multiplies.
constexpr int FKG = 64;              // 4-wide k groups per K quarter
constexpr int RPITCH = 128 + 4;      // floats per row of a wave's quarter product

// Wks[wave_id (128) = 4m + w][kg (64)][lane (64)] uint4: .xy / .zw = the B operands of column super groups 0 / 1: lane l holds,
// for gate column 64 sg + l of workgroup m (unit 32m + ((64 sg + l) >> 2), gate l & 3), the 4 weights of k = 256w + 4kg + e
__global__ __launch_bounds__(256) void pack_w_persist_ks_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 128L * FKG * 64) return;
    const int l = (int)(idx & 63);
    const int kg = (int)((idx >> 6) % FKG);
    const int wave_id = (int)(idx / (64L * FKG));
    const int m = wave_id >> 2, w = wave_id & 3;
    uint32_t o[4];
#pragma unroll
    for (int sg = 0; sg < 2; ++sg) {
        const int col = 64 * sg + l;
        const float* row = whh + ((long)(col & 3) * PH + 32 * m + (col >> 2)) * PH + 256 * w + 4 * kg;
        o[2 * sg] = lv_pack_bf16x2(row[0], row[1]);
        o[2 * sg + 1] = lv_pack_bf16x2(row[2], row[3]);
    }
    wpk[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

struct __attribute__((aligned(16))) FwdKsLds {
    uint32_t hl[PRMAX * HPITCH];        // gathered h_{t-1}: [row][k/2]; wave w writes and reads dwords [128w, 128w+128) of every row
    float red[2][4][PRMAX][RPITCH];     // [step parity][wave]: quarter product [row][gate column of the workgroup]
    int abort;
};

__global__ __launch_bounds__(256) void lstm_fwd_persist_ks_kernel(PersistFwdP p) {
    LV_BLOCK_SHARED(FwdKsLds, sm);
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);
    if (rows == 0) return;
    if (tid == 0) s_abort = 0;

    uint4 wreg[FKG];
    {
        const uint4* wp = p.wpk + (long)wave_id * FKG * 64 + l;
#pragma unroll
        for (int kg = 0; kg < FKG; ++kg) wreg[kg] = wp[kg * 64];
    }

    // this thread's (row, unit) pair: 8 rows x 32 units of the workgroup
    const int prow = tid >> 5, uw = tid & 31;
    const int punit = 32 * member + uw;
    const bool own = prow < rows;
    const long BH = (long)B * PH;
    const long pidx = (long)(b0 + (own ? prow : 0)) * PH + punit;
    float c_state = own ? p.cs[pidx] : 0.f;
    gran_t* const hx_g = p.hx + (long)group * 16 * (PH / 2);
    const long hx_par = (long)PGROUPS * 16 * (PH / 2);
    gran_t* const my_gran = hx_g + (long)prow * (PH / 2) + (punit >> 1);
    const bool publisher = own && !(uw & 1);

    {   // publish the initial state hs[0] as state 0 (tag 1)
        const uint32_t mine = lv_f32_to_bf16_bits(own ? p.hs[pidx] : 0.f);
        const uint32_t next = (uint32_t)__shfl_down((int)mine, 1, 64);
        if (publisher) gran_store(my_gran, ((gran_t)1u << 32) | (gran_t)(mine | (next << 16)));
    }

    float4 gxb[SB];
    float keepb[SB];
    float4 recb[SB];
    float cb[SB], hb[SB], hdb[SB];
    auto load_block = [&](int tb) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            gxb[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            keepb[s2] = 1.f;
            if (own && t < T) {
                gxb[s2] = *reinterpret_cast<const float4*>(p.gx + ((long)t * BH + pidx) * 4);
                if (p.hdrop && p.dmask) keepb[s2] = p.dmask[((long)(b0 + prow) * T + t) * PH + punit] ? p.dscale : 0.f;
            }
        }
    };
    auto store_block = [&](int tb) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            if (own && t < T) {
                lv_store_nt(f32x4{recb[s2].x, recb[s2].y, recb[s2].z, recb[s2].w},
                            reinterpret_cast<f32x4*>(p.gates + ((long)t * BH + pidx) * 4));
                lv_store_nt(cb[s2], p.cs + (long)(t + 1) * BH + pidx);
                lv_store_nt(hb[s2], p.hs + (long)(t + 1) * BH + pidx);
                if (p.hdrop) lv_store_nt(hdb[s2], p.hdrop + (long)t * BH + pidx);
            }
        }
    };
    load_block(0);
    __syncthreads();

    const int nq = rows * 128;                         // granules of this wave's K quarter
    for (int tb = 0; tb < T; tb += SB) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = tb + s2;
            if (t >= T) break;
            // ---- gather K-quarter w of state t (tag t+1) into this wave's part of the LDS image ---------------------------
            const gran_t* src = hx_g + (long)(t & 1) * hx_par + 128 * w;
            const uint32_t want = (uint32_t)(t + 1);
            for (int base = 0; base < nq; base += 64 * 8) {
                gran_t v[8];
                int spins = 0;
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int q = base + j * 64 + l;
                        const bool in = q < nq;
                        v[j] = gran_load(src + (in ? (q >> 7) * (PH / 2) + (q & 127) : 0));
                        ok = ok && (!in || (uint32_t)(v[j] >> 32) == want);
                    }
                    ok = __all(ok);
                    if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; break; }
                } while (!ok);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int q = base + j * 64 + l;
                    if (q < nq) sm.hl[(q >> 7) * HPITCH + 128 * w + (q & 127)] = (uint32_t)v[j];
                }
            }
            LV_WAIT_LDS();                             // the wave reads back only what its own lanes wrote

            // ---- this wave's quarter of the recurrent product: 4 batch rows per pass --------------------------------------
            float (*rd)[RPITCH] = sm.red[t & 1][w];
            for (int rb = 0; rb < rows; rb += 4) {
                const int ar = rb + (l & 3) < rows ? rb + (l & 3) : 0;       // rows beyond the slice re-read row 0: their D rows are unused
                const uint4* ap = reinterpret_cast<const uint4*>(sm.hl + ar * HPITCH + 128 * w);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
                uint4 abuf[2][8];                      // the lane's A row in 4 batches of 8 reads, batch c + 1 requested before batch c multiplies
#pragma unroll
                for (int u = 0; u < 8; ++u) abuf[0][u] = ap[u];
#pragma unroll
                for (int c = 0; c < FKG / 16; ++c) {
                    if (c + 1 < FKG / 16) {
#pragma unroll
                        for (int u = 0; u < 8; ++u) abuf[(c + 1) & 1][u] = ap[8 * (c + 1) + u];
                    }
                    LV_SCHED_BARRIER();
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const uint4 a = abuf[c & 1][u];
                        const uint4 b0_ = wreg[16 * c + 2 * u], b1_ = wreg[16 * c + 2 * u + 1];
                        acc0 = lv_mfma_4x4x4_16b_bf16(make_uint2(a.x, a.y), make_uint2(b0_.x, b0_.y), acc0);
                        acc1 = lv_mfma_4x4x4_16b_bf16(make_uint2(a.x, a.y), make_uint2(b0_.z, b0_.w), acc1);
                        acc0 = lv_mfma_4x4x4_16b_bf16(make_uint2(a.z, a.w), make_uint2(b1_.x, b1_.y), acc0);
                        acc1 = lv_mfma_4x4x4_16b_bf16(make_uint2(a.z, a.w), make_uint2(b1_.z, b1_.w), acc1);
                    }
                    LV_SCHED_BARRIER();
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { rd[rb + r][l] = acc0[r]; rd[rb + r][64 + l] = acc1[r]; }
            }
            __syncthreads();                           // the four quarter products (double-buffered by step parity)
            if (s_abort) { if (tid == 0) atomicExch(p.status, 100 + t); return; }

            // ---- epilogue: gates, cell update, hand-off of h_t ------------------------------------------------------------
            float h = 0.f;
            if (own) {
                const float4 q0 = *reinterpret_cast<const float4*>(&sm.red[t & 1][0][prow][4 * uw]);
                const float4 q1 = *reinterpret_cast<const float4*>(&sm.red[t & 1][1][prow][4 * uw]);
                const float4 q2 = *reinterpret_cast<const float4*>(&sm.red[t & 1][2][prow][4 * uw]);
                const float4 q3 = *reinterpret_cast<const float4*>(&sm.red[t & 1][3][prow][4 * uw]);
                const float ig = lv_sigmoid_fast(gxb[s2].x + ((q0.x + q1.x) + (q2.x + q3.x)));
                const float fg = lv_sigmoid_fast(gxb[s2].y + ((q0.y + q1.y) + (q2.y + q3.y)));
                const float gg = lv_tanh_fast(gxb[s2].z + ((q0.z + q1.z) + (q2.z + q3.z)));
                const float og = lv_sigmoid_fast(gxb[s2].w + ((q0.w + q1.w) + (q2.w + q3.w)));
                const float c = fg * c_state + ig * gg;
                h = og * lv_tanh_fast(c);
                c_state = c;
                recb[s2] = make_float4(ig, fg, gg, og);
                cb[s2] = c; hb[s2] = h; hdb[s2] = h * keepb[s2];
            }
            const uint32_t mine = lv_f32_to_bf16_bits(h);
            const uint32_t next = (uint32_t)__shfl_down((int)mine, 1, 64);
            if (publisher)
                gran_store(my_gran + (long)((t + 1) & 1) * hx_par, ((gran_t)(uint32_t)(t + 2) << 32) | (gran_t)(mine | (next << 16)));
        }
        store_block(tb);
        load_block(tb + SB);
    }
}

// =====================================================================================================================
// BPTT as one persistent launch.  Same decomposition (8 groups x 32 workgroups, a group owns a batch slice), mirrored:
// what travels between steps is dG[t] (4 gate gradients per unit), published by the lane that computed it as two 8-byte
// granules and gathered by every workgroup of the group into an LDS image [row][4H] in UNIT-major K order (n' = 4u + g).
// A workgroup owns 32 hidden units j; its 4 waves split the contraction over n' into quarters and each keeps its
// 1024 x 32 slice of W_hh in registers (64 KB, as in the forward kernel).  The quarter products are summed through LDS,
// and since the workgroup now holds dh_{t-1} for its units COMPLETELY, the gate-gradient math of step t-1 runs right
// there: one phase per timestep instead of two launches (elementwise + split-K matmul).  Bulk I/O in SB-step blocks
// as in the forward kernel.  H == 1024, at most 4 batch rows per group (B <= 32).
constexpr int BR = 4;                            // batch rows per group (LDS image: BR x 4H bf16)
constexpr int GPITCH = 2 * PH + 16;              // dwords per image row (4H bf16 = 2H dwords, + bank-spreading pad)

// Wpb[wave_id (128)][ks (32)][nb (2)][lane (64)]: wave (m, w) = workgroup m's K-quarter w; lane (c, kq) holds, for column
// unit j = 32m + 16nb + c, the 8 weights W_hh[gate*H + unit][j] of n' = 1024w + 32ks + 8kq + e (unit = n' >> 2, gate = n' & 3)
__global__ __launch_bounds__(256) void pack_w_persist_bwd_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 128L * PKS * 2 * 64) return;
    const int l = (int)(idx & 63);
    const int nb = (int)((idx >> 6) & 1);
    const int ks = (int)((idx >> 7) % PKS);
    const int wave_id = (int)(idx / (64L * 2 * PKS));
    const int m = wave_id >> 2, w = wave_id & 3;
    const int c = l & 15, kq = l >> 4;
    const int j = 32 * m + 16 * nb + c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int np = 1024 * w + 32 * ks + 8 * kq + e;
        v[e] = whh[((long)(np & 3) * PH + (np >> 2)) * PH + j];
    }
    wpk[idx] = make_uint4(lv_pack_bf16x2(v[0], v[1]), lv_pack_bf16x2(v[2], v[3]), lv_pack_bf16x2(v[4], v[5]), lv_pack_bf16x2(v[6], v[7]));
}

struct PersistBwdP {
    const float* dh_ext; const float* dh_last; const uint8_t* dmask; float dscale;
    const uint4* wpk;
    const float* gates; const float* cs; const float* hs;
    uint16_t* dG16; float* dGsum;
    float* dh0; float* dc0; int tanh_init;
    gran_t* gxch;               // exchange: [2 parity][8 groups][BR rows][2H] granules, zeroed before the launch
    int* status;
    int T, B, R;
};

struct __attribute__((aligned(16))) BwdLds {
    uint32_t gl[BR * GPITCH];           // gathered dG[t+1], [row][n'/2]
    uint16_t og[SB][BR][4][32];         // dG of one I/O block: [step][row][gate][unit in WG]
    float red[2][4][16][33];            // [phase parity][wave]: quarter product [row][unit]
    int abort;
};
__global__ __launch_bounds__(256) void lstm_bwd_persist_kernel(PersistBwdP p) {
    LV_BLOCK_SHARED(BwdLds, sm);
    uint32_t (&gl)[BR * GPITCH] = sm.gl;
    float (&red)[2][4][16][33] = sm.red;
    uint16_t (&og)[SB][BR][4][32] = sm.og;
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);
    if (rows == 0) return;
    if (tid == 0) s_abort = 0;

    uint4 wreg[PKS][2];
    {
        const uint4* wp = p.wpk + (long)wave_id * PKS * 2 * 64 + l;
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) wreg[ks][nb] = wp[(ks * 2 + nb) * 64];
    }

    // this lane's (row, unit) pair: lanes 0..31 of each wave, unit = 32*member + 8*w + (l & 7)
    const int prow = (l >> 3) & 3, ul = l & 7;
    const int uw = 8 * w + ul;                                               // unit within the workgroup
    const int punit = 32 * member + uw;
    const bool own = l < 32 && prow < rows;
    const long BH = (long)B * PH;
    const long pidx = (long)(b0 + (own ? prow : 0)) * PH + punit;
    gran_t* const gx_g = p.gxch + (long)group * BR * (2 * PH);
    const long gx_par = (long)PGROUPS * BR * (2 * PH);
    gran_t* const my_gran = gx_g + (long)prow * (2 * PH) + 2 * punit;       // granules (i,f) and (g,o) of this pair

    float dc_rec = 0.f;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};

    float dhb[SB], keepb[SB], ctb[SB + 1];
    float4 recb[SB];
    uint32_t outb[SB][2];            // bf16 pairs (i,f), (g,o) of the block's steps, stored at the block boundary
    auto load_block = [&](int t_hi) {            // steps t_hi, t_hi-1, ..., t_hi-SB+1
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = t_hi - s2;
            dhb[s2] = 0.f; keepb[s2] = 1.f; recb[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (own && t >= 0) {
                if (p.dh_ext) dhb[s2] = p.dh_ext[(long)t * BH + pidx];
                if (p.dh_ext && p.dmask) keepb[s2] = p.dmask[((long)(b0 + prow) * T + t) * PH + punit] ? p.dscale : 0.f;
                recb[s2] = *reinterpret_cast<const float4*>(p.gates + ((long)t * BH + pidx) * 4);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 <= SB; ++s2) {        // ctb[s2] = c of step t_hi - s2 (= cs[t+1]); ctb[s2+1] is that step's c_{t-1}
            const int t = t_hi - s2;
            ctb[s2] = (own && t + 1 >= 0) ? p.cs[(long)(t + 1) * BH + pidx] : 0.f;
        }
    };
    // The dG image is gate-major ([t][b][g*H + u]: the contraction index of the weight-gradient GEMMs), a lane holds the four
    // gates of ONE unit: written directly that is four 2-byte stores per step, each a partial-line write (measured by PMC:
    // 262 MB written per launch for a 52 MB image).  The block's values are transposed through LDS instead, so that the
    // workgroup writes (step, row, gate) segments of 32 units = 64 B as 16-byte streaming stores: 2 per thread per block.
    auto store_block = [&](int t_hi) {
        if (own) {
#pragma unroll
            for (int s2 = 0; s2 < SB; ++s2) {
                og[s2][prow][0][uw] = (uint16_t)(outb[s2][0] & 0xFFFFu);
                og[s2][prow][1][uw] = (uint16_t)(outb[s2][0] >> 16);
                og[s2][prow][2][uw] = (uint16_t)(outb[s2][1] & 0xFFFFu);
                og[s2][prow][3][uw] = (uint16_t)(outb[s2][1] >> 16);
            }
        }
        __syncthreads();
        for (int c = tid; c < SB * BR * 4 * 4; c += 256) {      // 16-byte chunks: [step][row][gate][quarter of 32 units]
            const int q = c & 3, g = (c >> 2) & 3, r = (c >> 4) & 3, s2 = c >> 6;
            const int t = t_hi - s2;
            if (t >= 0 && r < rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&og[s2][r][g][8 * q]);      // 8 bf16, moved as raw bits
                uint16_t* dst = p.dG16 + ((long)t * B + (b0 + r)) * 4 * PH + (long)g * PH + 32 * member + 8 * q;
                lv_store_nt(v, reinterpret_cast<f32x4*>(dst));
            }
        }
        __syncthreads();                                        // og is rewritten at the next block boundary
    };
    load_block(T - 1);
    __syncthreads();

    const int arow = (l & 15) < rows ? (l & 15) : 0, kq = l >> 4;

    // one recurrent phase: gather the image tagged `want`, contract it with this workgroup's columns, leave the summed
    // dh for (row, unit) pairs in `dh_rec` of the owning lanes.  Returns false on a hand-off timeout.
    // Wave w contracts over K-quarter w only, so it gathers exactly that quarter of every row (granules
    // [row][512w, 512w+512)) into its own part of the LDS image and starts its MFMAs without waiting for the other
    // waves; the one workgroup barrier of the phase is the exchange of the four quarter products.
    const int nq = rows * 512;                         // granules this wave gathers per phase
    int phase = 0;
    auto recurrent = [&](uint32_t want, float& dh_rec) -> bool {
        const gran_t* src = gx_g + (long)(want & 1) * gx_par;
        constexpr int GJ = 16;                         // granules per lane per polling round (all in flight together)
        bool fine = true;
        for (int base = 0; base < nq && fine; base += 64 * GJ) {
            gran_t v[GJ];
            int spins = 0;
            bool ok;
            do {
                ok = true;
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int q = base + j * 64 + l;
                    const bool in = q < nq;
                    v[j] = gran_load(src + (in ? (q >> 9) * (2 * PH) + 512 * w + (q & 511) : 0));
                    ok = ok && (!in || (uint32_t)(v[j] >> 32) == want);
                }
                ok = __all(ok);
                if (!ok && ++spins > SPIN_LIMIT) { fine = false; break; }
            } while (!ok);
#pragma unroll
            for (int j = 0; j < GJ; ++j) {
                const int q = base + j * 64 + l;
                if (q < nq) gl[(q >> 9) * GPITCH + 512 * w + (q & 511)] = (uint32_t)v[j];
            }
        }
        if (!fine) s_abort = 1;
        LV_WAIT_LDS();            // lgkmcnt(0): the wave reads back what its own lanes just wrote
        f32x4 acc[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        const uint4* arowp = reinterpret_cast<const uint4*>(gl + arow * GPITCH + 512 * w) + kq;     // K-quarter w: 1024 n' = 512 dwords
#pragma unroll
        for (int ks = 0; ks < PKS; ++ks) {
            const uint4 a = arowp[ks * 4];
            acc[(ks & 1) * 2 + 0] = lv_mfma_16x16x32_bf16(a, wreg[ks][0], acc[(ks & 1) * 2 + 0]);
            acc[(ks & 1) * 2 + 1] = lv_mfma_16x16x32_bf16(a, wreg[ks][1], acc[(ks & 1) * 2 + 1]);
        }
        float (*rd)[16][33] = red[phase & 1];
        phase ^= 1;
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) rd[w][(l >> 4) * 4 + r][nb * 16 + (l & 15)] = acc[nb][r] + acc[2 + nb][r];
        __syncthreads();                               // quarter products of all 4 waves (double-buffered by phase parity)
        if (s_abort) return false;
        dh_rec = (rd[0][prow][uw] + rd[1][prow][uw]) + (rd[2][prow][uw] + rd[3][prow][uw]);
        return true;
    };

    for (int t_hi = T - 1; t_hi >= 0; t_hi -= SB) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = t_hi - s2;
            if (t < 0) break;
            float dh_rec = 0.f;
            if (t < T - 1) {
                if (!recurrent((uint32_t)(T - 1 - t), dh_rec)) { if (tid == 0) atomicExch(p.status, 200 + t); return; }
            }
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            if (own) {
                float dh = dhb[s2] * keepb[s2] + dh_rec;
                if (t == T - 1 && p.dh_last) dh += p.dh_last[pidx];
                const float ig = recb[s2].x, fg = recb[s2].y, gg = recb[s2].z, og = recb[s2].w;
                const float tc = lv_tanh_fast(ctb[s2]);
                const float dc = dh * og * (1.f - tc * tc) + dc_rec;
                const float d_o = dh * tc;
                const float d_i = dc * gg, d_g = dc * ig, d_f = dc * ctb[s2 + 1];
                da[0] = d_i * ig * (1.f - ig);
                da[1] = d_f * fg * (1.f - fg);
                da[2] = d_g * (1.f - gg * gg);
                da[3] = d_o * og * (1.f - og);
                dc_rec = dc * fg;
#pragma unroll
                for (int g = 0; g < 4; ++g) gsum[g] += da[g];
            }
            const uint32_t lo = lv_pack_bf16x2(da[0], da[1]), hi = lv_pack_bf16x2(da[2], da[3]);
            outb[s2][0] = lo; outb[s2][1] = hi;
            if (own) {      // hand dG[t] to the group: tag T - t, parity (T - t) & 1
                gran_t* dst = my_gran + (long)((T - t) & 1) * gx_par;
                const gran_t tag = (gran_t)(uint32_t)(T - t) << 32;
                gran_store(dst, tag | (gran_t)lo);
                gran_store(dst + 1, tag | (gran_t)hi);
            }
        }
        store_block(t_hi);
        load_block(t_hi - SB);
    }

    // closing phase: dh0 = dG[0] . W_hh, dc0 = dc_rec (+ dh0 * (1 - h0^2) when h0 = tanh(c0))
    if (own) {
#pragma unroll
        for (int g = 0; g < 4; ++g) p.dGsum[(long)(b0 + prow) * 4 * PH + (long)g * PH + punit] = gsum[g];
    }
    if (p.dh0 || p.tanh_init) {
        float s0 = 0.f;
        if (!recurrent((uint32_t)T, s0)) { if (tid == 0) atomicExch(p.status, 300); return; }
        if (own) {
            if (p.dh0) p.dh0[pidx] = s0;
            float dc = dc_rec;
            if (p.tanh_init) { const float h0 = p.hs[pidx]; dc += s0 * (1.f - h0 * h0); }
            if (p.dc0) p.dc0[pidx] = dc;
        }
    } else if (own && p.dc0) {
        p.dc0[pidx] = dc_rec;
    }
}

// =====================================================================================================================
// BPTT, reduce-scatter form.  The kernel above all-gathers dG[t] (R x 4H values) into every workgroup: 8192 granules per
// workgroup and timestep at R = 4, four times the forward's hand-off, and the step costs 5.1-5.6 us against 3.2-3.8 us.
// Here a workgroup keeps the dG of its OWN 32 units (it computes them), multiplies them with its 128 gate ROWS of W_hh --
// a partial sum of dh_{t-1} for ALL 1024 units -- and sends every other workgroup the 32-unit slice that workgroup owns:
// 32 x R f32 partials per (sender, receiver) pair, two per 8-byte granule as 28-bit floats (sign, exponent, 19 mantissa
// bits: 1e-6 relative, three orders below the bf16 rounding of the operands) next to an 8-bit phase tag.  A workgroup then
// receives 32 senders x 64 granules = 2048 granules per timestep -- the forward's count -- and sums them in a fixed order.
// Same decomposition otherwise: 8 groups x 32 workgroups, a group owns R <= 4 batch rows, W_hh slice in registers (wave w
// holds its workgroup's 128 k-rows x output columns [256w, 256w+256)), bulk I/O in SB-step blocks, bounded spins.
//   per timestep: poll 8 granules per lane (one round) -> sum over senders (registers + 2 shuffles: no barrier) -> gate
//   gradients -> dG image of the workgroup in LDS -> ONE barrier -> 128 v_mfma_f32_4x4x4_16b_bf16 per wave (the batch slice IS
//   4 rows: the 16-block MFMA does the work of a 16-row tile's useful quarter in half the pipe time, and its D layout -- all
//   4 rows of one output column in one lane -- is the granule layout) -> 8 granule stores per lane straight from the
//   accumulators (each wave sends to the 8 workgroups that own its 256 columns; a wave store is two 256-byte runs).
constexpr int RS_KG = 32;                        // 4-wide k groups over the workgroup's 128 gate rows
constexpr int RS_SG = 4;                         // 64-column super groups per wave (16 blocks x 4 columns per MFMA)
constexpr int DPITCH = 64 + 4;                   // dwords per row of the dG image (128 bf16 + pad: rows 4 bank groups apart)

// Wrs[wave_id (128) = 4m + w][kg (32)][sgp (2)][lane (64)] uint4: lane l holds, for output columns j = 256w + 64(2 sgp + h) + l
// (h = 0, 1: .xy / .zw), the 4 weights W_hh[gate*H + unit][j] of the local gate rows n'' = 4kg + e (unit = 32m + kg, gate = e)
__global__ __launch_bounds__(256) void pack_w_persist_bwd_rs_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= 128L * RS_KG * 2 * 64) return;
    const int l = (int)(idx & 63);
    const int sgp = (int)((idx >> 6) & 1);
    const int kg = (int)((idx >> 7) % RS_KG);
    const int wave_id = (int)(idx / (64L * 2 * RS_KG));
    const int m = wave_id >> 2, w = wave_id & 3;
    uint32_t o[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int j = 256 * w + 64 * (2 * sgp + h) + l;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = whh[((long)e * PH + 32 * m + kg) * PH + j];
        o[2 * h] = lv_pack_bf16x2(v[0], v[1]);
        o[2 * h + 1] = lv_pack_bf16x2(v[2], v[3]);
    }
    wpk[idx] = make_uint4(o[0], o[1], o[2], o[3]);
}

struct __attribute__((aligned(16))) BwdRsLds {
    uint32_t dgl[2][16 * DPITCH];       // [step parity] dG of this workgroup's 128 gate rows as the MFMA A image, rows < R valid
    uint16_t og[SB][BR][4][32];         // dG of one I/O block: [step][row][gate][unit in WG]
    int abort;
};

__global__ __launch_bounds__(256) void lstm_bwd_persist_rs_kernel(PersistBwdP p) {
    LV_BLOCK_SHARED(BwdRsLds, sm);
    uint16_t (&og)[SB][BR][4][32] = sm.og;
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);
    if (rows == 0) return;
    if (tid == 0) s_abort = 0;

    uint4 wreg[RS_KG][2];               // [k group][super-group pair]: .xy / .zw = the B operands of super groups 2 sgp, 2 sgp + 1
    {
        const uint4* wp = p.wpk + (long)wave_id * RS_KG * 2 * 64 + l;
#pragma unroll
        for (int kg = 0; kg < RS_KG; ++kg)
#pragma unroll
            for (int sgp = 0; sgp < 2; ++sgp) wreg[kg][sgp] = wp[(kg * 2 + sgp) * 64];
    }

    // this lane's (row, unit) pair: lanes 0..31 of each wave; wave w receives the sums of rows {2(w>>1), 2(w>>1)+1} x units
    // [16(w&1), 16(w&1)+16) of the workgroup -- exactly the pairs its first 32 lanes own
    const int prow = 2 * (w >> 1) + ((l >> 4) & 1), uw = 16 * (w & 1) + (l & 15);
    const int punit = 32 * member + uw;
    const bool own = l < 32 && prow < rows;
    const long BH = (long)B * PH;
    const long pidx = (long)(b0 + (own ? prow : 0)) * PH + punit;
    // exchange: [parity][group][receiver (32)][sender (32)][64 granules: (row pair rp, unit u) at rp*32 + u]
    const long px_par = (long)PGROUPS * PMEMBERS * PMEMBERS * 64;
    gran_t* const px_g = p.gxch + (long)group * PMEMBERS * PMEMBERS * 64;
    const gran_t* const rx = px_g + (long)member * PMEMBERS * 64 + (8 * (l >> 4)) * 64 + 16 * w + (l & 15);   // + j*64 per sender
    // lane l of super group sg holds output column 64 sg + l: receiver 8w + 2sg + (l >> 5), its unit l & 31
    gran_t* const tx = px_g + ((long)(8 * w + (l >> 5)) * PMEMBERS + member) * 64 + (l & 31);                 // + sg*2*PMEMBERS*64, + 32 for rows 2,3

    float dc_rec = 0.f;
    float gsum[4] = {0.f, 0.f, 0.f, 0.f};
    float dhb[SB], keepb[SB], ctb[SB + 1];
    float4 recb[SB];
    uint32_t outb[SB][2];
    auto load_block = [&](int t_hi) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = t_hi - s2;
            dhb[s2] = 0.f; keepb[s2] = 1.f; recb[s2] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (own && t >= 0) {
                if (p.dh_ext) dhb[s2] = p.dh_ext[(long)t * BH + pidx];
                if (p.dh_ext && p.dmask) keepb[s2] = p.dmask[((long)(b0 + prow) * T + t) * PH + punit] ? p.dscale : 0.f;
                recb[s2] = *reinterpret_cast<const float4*>(p.gates + ((long)t * BH + pidx) * 4);
            }
        }
#pragma unroll
        for (int s2 = 0; s2 <= SB; ++s2) {
            const int t = t_hi - s2;
            ctb[s2] = (own && t + 1 >= 0) ? p.cs[(long)(t + 1) * BH + pidx] : 0.f;
        }
    };
    auto store_block = [&](int t_hi) {
        if (own) {
#pragma unroll
            for (int s2 = 0; s2 < SB; ++s2) {
                og[s2][prow][0][uw] = (uint16_t)(outb[s2][0] & 0xFFFFu);
                og[s2][prow][1][uw] = (uint16_t)(outb[s2][0] >> 16);
                og[s2][prow][2][uw] = (uint16_t)(outb[s2][1] & 0xFFFFu);
                og[s2][prow][3][uw] = (uint16_t)(outb[s2][1] >> 16);
            }
        }
        __syncthreads();
        for (int c = tid; c < SB * BR * 4 * 4; c += 256) {      // 16-byte chunks: [step][row][gate][quarter of 32 units]
            const int q = c & 3, g = (c >> 2) & 3, r = (c >> 4) & 3, s2 = c >> 6;
            const int t = t_hi - s2;
            if (t >= 0 && r < rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&og[s2][r][g][8 * q]);
                uint16_t* dst = p.dG16 + ((long)t * B + (b0 + r)) * 4 * PH + (long)g * PH + 32 * member + 8 * q;
                lv_store_nt(v, reinterpret_cast<f32x4*>(dst));
            }
        }
        __syncthreads();
    };
    load_block(T - 1);
    __syncthreads();

    const int arow = (l & 3) < rows ? (l & 3) : 0;      // A row of this lane; rows beyond the slice re-read row 0 (their D rows are ignored)
    const bool closing = p.dh0 || p.tanh_init;

    // receive phase k: the 32 senders' partial sums of dh for this workgroup's units; lanes < 32 get their pair's total
    auto receive = [&](int k, float& dh_rec) -> bool {
        const gran_t* src = rx + (long)(k & 1) * px_par;
        const uint32_t want = rs_tag(k);
        gran_t v[8];
        int spins = 0;
        bool ok;
        do {
            ok = true;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = gran_load(src + j * 64);
                ok = ok && (uint32_t)(v[j] >> 56) == want;
            }
            ok = __all(ok);
            if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; return false; }
        } while (!ok);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { a += rs_lo(v[j]); b += rs_hi(v[j]); }
        a += __shfl_xor(a, 16, 64); b += __shfl_xor(b, 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        dh_rec = (l & 16) ? b : a;
        return true;
    };
    // multiply the dG image of step parity `par` with this wave's columns and send phase k to their owners
    auto send = [&](int par, int k) {
        const uint32_t* img = sm.dgl[par] + arow * DPITCH;
        uint4 afr[RS_KG / 2];            // this lane's A row, all 128 k: 16 broadcast reads
#pragma unroll
        for (int q = 0; q < RS_KG / 2; ++q) afr[q] = *reinterpret_cast<const uint4*>(img + 4 * q);
        const uint32_t tag = rs_tag(k);
        gran_t* dst = tx + (long)(k & 1) * px_par;
#pragma unroll
        for (int sgp = 0; sgp < 2; ++sgp) {
            f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < RS_KG / 2; ++q) {
                acc0 = lv_mfma_4x4x4_16b_bf16(make_uint2(afr[q].x, afr[q].y), make_uint2(wreg[2 * q][sgp].x, wreg[2 * q][sgp].y), acc0);
                acc1 = lv_mfma_4x4x4_16b_bf16(make_uint2(afr[q].x, afr[q].y), make_uint2(wreg[2 * q][sgp].z, wreg[2 * q][sgp].w), acc1);
                acc0 = lv_mfma_4x4x4_16b_bf16(make_uint2(afr[q].z, afr[q].w), make_uint2(wreg[2 * q + 1][sgp].x, wreg[2 * q + 1][sgp].y), acc0);
                acc1 = lv_mfma_4x4x4_16b_bf16(make_uint2(afr[q].z, afr[q].w), make_uint2(wreg[2 * q + 1][sgp].z, wreg[2 * q + 1][sgp].w), acc1);
            }
            // the two super groups' columns go out while the next pair multiplies
            gran_t* d0 = dst + (long)(2 * sgp) * 2 * PMEMBERS * 64;
            gran_t* d1 = d0 + 2L * PMEMBERS * 64;
            gran_store(d0, rs_pack(acc0[0], acc0[1], tag));
            gran_store(d0 + 32, rs_pack(acc0[2], acc0[3], tag));
            gran_store(d1, rs_pack(acc1[0], acc1[1], tag));
            gran_store(d1 + 32, rs_pack(acc1[2], acc1[3], tag));
        }
    };

    for (int t_hi = T - 1; t_hi >= 0; t_hi -= SB) {
#pragma unroll
        for (int s2 = 0; s2 < SB; ++s2) {
            const int t = t_hi - s2;
            if (t < 0) break;
            float dh_rec = 0.f;
            if (t < T - 1) receive(T - 1 - t, dh_rec);        // on a timeout s_abort is set: everybody leaves after the barrier below
            float da[4] = {0.f, 0.f, 0.f, 0.f};
            if (own) {
                float dh = dhb[s2] * keepb[s2] + dh_rec;
                if (t == T - 1 && p.dh_last) dh += p.dh_last[pidx];
                const float ig = recb[s2].x, fg = recb[s2].y, gg = recb[s2].z, og_ = recb[s2].w;
                const float tc = lv_tanh_fast(ctb[s2]);
                const float dc = dh * og_ * (1.f - tc * tc) + dc_rec;
                const float d_o = dh * tc;
                const float d_i = dc * gg, d_g = dc * ig, d_f = dc * ctb[s2 + 1];
                da[0] = d_i * ig * (1.f - ig);
                da[1] = d_f * fg * (1.f - fg);
                da[2] = d_g * (1.f - gg * gg);
                da[3] = d_o * og_ * (1.f - og_);
                dc_rec = dc * fg;
#pragma unroll
                for (int g = 0; g < 4; ++g) gsum[g] += da[g];
            }
            const uint32_t lo = lv_pack_bf16x2(da[0], da[1]), hi = lv_pack_bf16x2(da[2], da[3]);
            outb[s2][0] = lo; outb[s2][1] = hi;
            const int par = (T - t) & 1;
            if (own) { sm.dgl[par][prow * DPITCH + 2 * uw] = lo; sm.dgl[par][prow * DPITCH + 2 * uw + 1] = hi; }
            __syncthreads();                    // the workgroup's dG image of this step (double-buffered by step parity)
            if (s_abort) break;
            if (t > 0 || closing) send(par, T - t);
        }
        if (s_abort) { if (tid == 0) atomicExch(p.status, 200 + (t_hi < 0 ? 0 : t_hi)); return; }
        store_block(t_hi);
        load_block(t_hi - SB);
    }

    if (own) {
#pragma unroll
        for (int g = 0; g < 4; ++g) p.dGsum[(long)(b0 + prow) * 4 * PH + (long)g * PH + punit] = gsum[g];
    }
    if (closing) {
        float s0 = 0.f;
        const bool fine = receive(T, s0);
        __syncthreads();
        if (!fine || s_abort) { if (tid == 0) atomicExch(p.status, 300); return; }
        if (own) {
            if (p.dh0) p.dh0[pidx] = s0;
            float dc = dc_rec;
            if (p.tanh_init) { const float h0 = p.hs[pidx]; dc += s0 * (1.f - h0 * h0); }
            if (p.dc0) p.dc0[pidx] = dc;
        }
    } else if (own && p.dc0) {
        p.dc0[pidx] = dc_rec;
    }
}

constexpr long WPK_BYTES = 128L * PKS * 2 * 64 * 16;                           // one packed bf16 image of W_hh (8 MB)
constexpr long XCH_FWD_BYTES = 2L * PGROUPS * 16 * (PH / 2) * 8;                 // h exchange, two parities
constexpr long XCH_BWD_BYTES = 2L * PGROUPS * BR * (2 * PH) * 8;                 // dG exchange, two parities
constexpr long XCH_RS_BYTES = 2L * PGROUPS * PMEMBERS * PMEMBERS * 64 * 8;       // partial-sum exchange (reduce-scatter BPTT), two parities

}  // namespace

extern "C" long lv_lstm_persist_wpk_floats(void) { return WPK_BYTES / 4; }
extern "C" long lv_lstm_persist_xch_floats(void) { return XCH_RS_BYTES / 4 + 64; }      // the largest of the three exchanges

// W_hh [4H][H] f32 -> the register image of the forward (backward = 0; 3 = its K-split form), all-gather BPTT (1) or
// reduce-scatter BPTT (2) kernel:
// lv_lstm_persist_wpk_floats() floats, 16-byte aligned.  Re-run only when the weights change.
extern "C" int lv_lstm_persist_pack(const float* whh, float* wpk, int backward, int H, void* stream) {
    if (!whh || !wpk) return LV_ERR_ARG;
    if (H != PH) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0) return LV_ERR_ALIGN;
    const dim3 grid((unsigned)lv_cdiv(128L * PKS * 2 * 64, 256)), block(256);      // every image: 128 x 32 x 2 x 64 uint4
    if (backward == 3) LV_LAUNCH(pack_w_persist_ks_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk));
    else if (backward == 2) LV_LAUNCH(pack_w_persist_bwd_rs_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk));
    else if (backward) LV_LAUNCH(pack_w_persist_bwd_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk));
    else LV_LAUNCH(pack_w_persist_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk));
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// BPTT in one persistent launch.  Arguments as lv_lstm_bwd_bf16_img with W_hh replaced by its packed image
// (lv_lstm_persist_pack(..., backward = 1)), an exchange buffer of lv_lstm_persist_xch_floats() floats and a device status
// word.  LV_ERR_UNSUPPORTED unless H == 1024, B <= 32 and the device has >= 256 CUs.
extern "C" int lv_lstm_bwd_bf16_persist(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                                        const float* wpk, const float* gates, const float* hs, const float* cs,
                                        float* dG, uint16_t* dG16, float* dGsum, float* xch, int* status, float* dh0, float* dc0,
                                        int tanh_init, int T, int B, int H, void* stream) {
    if (!wpk || !gates || !cs || (!dG && !dG16) || !dGsum || !xch || !status) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (tanh_init && !hs) return LV_ERR_ARG;
    if (H != PH || B > BR * PGROUPS || dG || !dG16) return LV_ERR_UNSUPPORTED;   // image-only: the f32 dG copy is not produced
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)gates) & 15) != 0 || (((uintptr_t)xch) & 15) != 0 || (((uintptr_t)dG16) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;
    gran_t* gxch = reinterpret_cast<gran_t*>(xch);
    hipMemsetAsync(gxch, 0, (size_t)XCH_BWD_BYTES, (hipStream_t)stream);
    const int R = (B + PGROUPS - 1) / PGROUPS;
    PersistBwdP p{dh_ext, dh_last, dmask, dscale, reinterpret_cast<const uint4*>(wpk), gates, cs, hs, dG16, dGsum, dh0, dc0, tanh_init,
                  gxch, status, T, B, R};
    LV_LAUNCH_RESIDENT(lstm_bwd_persist_kernel, dim3(PGROUPS * PMEMBERS), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same BPTT in its reduce-scatter form (lstm_bwd_persist_rs_kernel; weights packed with backward = 2): a quarter of the
// hand-off granules per timestep.  Same arguments, same outputs up to f32 summation order.
extern "C" int lv_lstm_bwd_bf16_persist_rs(const float* dh_ext, const float* dh_last, const uint8_t* dmask, float dscale,
                                           const float* wpk, const float* gates, const float* hs, const float* cs,
                                           float* dG, uint16_t* dG16, float* dGsum, float* xch, int* status, float* dh0, float* dc0,
                                           int tanh_init, int T, int B, int H, void* stream) {
    if (!wpk || !gates || !cs || (!dG && !dG16) || !dGsum || !xch || !status) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (tanh_init && !hs) return LV_ERR_ARG;
    if (H != PH || B > BR * PGROUPS || dG || !dG16) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)gates) & 15) != 0 || (((uintptr_t)xch) & 15) != 0 || (((uintptr_t)dG16) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;
    gran_t* gxch = reinterpret_cast<gran_t*>(xch);
    hipMemsetAsync(gxch, 0, (size_t)XCH_RS_BYTES, (hipStream_t)stream);
    const int R = (B + PGROUPS - 1) / PGROUPS;
    PersistBwdP p{dh_ext, dh_last, dmask, dscale, reinterpret_cast<const uint4*>(wpk), gates, cs, hs, dG16, dGsum, dh0, dc0, tanh_init,
                  gxch, status, T, B, R};
    LV_LAUNCH_RESIDENT(lstm_bwd_persist_rs_kernel, dim3(PGROUPS * PMEMBERS), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Forward recurrence in one persistent launch.  Arguments as lv_lstm_fwd_bf16_ug (gx unit-major) with W_hh replaced by
// its packed image (lv_lstm_persist_pack(..., backward = 0)), an exchange buffer and a device status word (0 = ok;
// written non-zero if a hand-off timed out).
extern "C" int lv_lstm_fwd_bf16_persist(const float* gx, const float* wpk, float* hs, float* cs, float* gates,
                                        const uint8_t* dmask, float dscale, float* hdrop, float* xch, int* status,
                                        int T, int B, int H, void* stream) {
    if (!gx || !wpk || !hs || !cs || !gates || !xch || !status) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (dmask && !hdrop) return LV_ERR_ARG;
    if (H != PH || B > PRMAX * PGROUPS) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)gates) & 15) != 0 || (((uintptr_t)gx) & 15) != 0 || (((uintptr_t)xch) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;      // all 256 workgroups must be resident at once
    if (T == 0) return LV_OK;
    gran_t* hx = reinterpret_cast<gran_t*>(xch);
    hipMemsetAsync(hx, 0, (size_t)XCH_FWD_BYTES, (hipStream_t)stream);
    const int R = (B + PGROUPS - 1) / PGROUPS;
    PersistFwdP p{gx, reinterpret_cast<const uint4*>(wpk), hs, cs, gates, dmask, dscale, hdrop, hx, status, T, B, R};
    LV_LAUNCH_RESIDENT(lstm_fwd_persist_kernel, dim3(PGROUPS * PMEMBERS), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same forward recurrence in its K-split form (weights packed with backward = 3; red LDS tile needs a 16-byte aligned gx).
// Forward recurrence in one persistent launch.  Arguments as lv_lstm_fwd_bf16_ug (gx unit-major) with W_hh replaced by
// its packed image (lv_lstm_persist_pack(..., backward = 0)), an exchange buffer and a device status word (0 = ok;
// written non-zero if a hand-off timed out).
extern "C" int lv_lstm_fwd_bf16_persist_ks(const float* gx, const float* wpk, float* hs, float* cs, float* gates,
                                        const uint8_t* dmask, float dscale, float* hdrop, float* xch, int* status,
                                        int T, int B, int H, void* stream) {
    if (!gx || !wpk || !hs || !cs || !gates || !xch || !status) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (dmask && !hdrop) return LV_ERR_ARG;
    if (H != PH || B > PRMAX * PGROUPS) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)gates) & 15) != 0 || (((uintptr_t)gx) & 15) != 0 || (((uintptr_t)xch) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;      // all 256 workgroups must be resident at once
    if (T == 0) return LV_OK;
    gran_t* hx = reinterpret_cast<gran_t*>(xch);
    hipMemsetAsync(hx, 0, (size_t)XCH_FWD_BYTES, (hipStream_t)stream);
    const int R = (B + PGROUPS - 1) / PGROUPS;
    PersistFwdP p{gx, reinterpret_cast<const uint4*>(wpk), hs, cs, gates, dmask, dscale, hdrop, hx, status, T, B, R};
    LV_LAUNCH_RESIDENT(lstm_fwd_persist_ks_kernel, dim3(PGROUPS * PMEMBERS), dim3(256), 0, stream, p);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
