// lv_lstm_persist16.hip -- the persistent LSTM recurrences for UP TO 16 BATCH ROWS PER XCD GROUP (bf16 recurrent operands, H = 1024).
//
// The rounds 1-2 kernels (lv_lstm_persist.hip, retired in round 4: these are faster at every row count, profiles/r03z_persist16_probe.txt)
// carried 4 rows per group (B <= 32 on 8 groups) on the 4x4x4 MFMA.  Two things need more rows per group: the stress
// configuration (B = 128 per GPU: 16 rows on each of the 8 groups) and running a B = 32 recurrence on HALF the chip (4 groups x 8
// rows).  Decomposition and hand-off: a group = the 32 workgroups with the same blockIdx % 8; a group owns a slice of the batch
// and carries it through all T steps, groups never talk; W_hh stays register-resident for the whole call (a wave's 64 KB slice
// in 256 AGPRs); per timestep the only exchange is h_t (forward, all-gather) or partial dh sums (BPTT, reduce-scatter) inside
// the group, as tagged granules gathered by polling; bulk I/O in blocks of timesteps; all spins bounded.  Since the end of round 6 a
// granule is 16 bytes and every dword of it carries its own tag (LV_FWD_Q / LV_RS_Q below; the 8-byte granules with one tag per
// granule remain as the A/B builds): the hand-off is bound by the NUMBER of load / store instructions, not by their bytes, and
// 16-byte atomicity is assumed nowhere (the CI emulator tears the stores).  The contraction:
//
//   * v_mfma_f32_16x16x32_bf16 with the operands SWAPPED: the weights are the A operand (16 gate columns / output units as the
//     tile's rows), the batch rows are the B operand's 16 columns.  The matrix-pipe time is the one the 4-row kernels already pay
//     (64 MFMAs x 16 cycles per wave and timestep) for up to 16 rows, and the D tile comes out as [gate column][batch row]: a
//     lane holds FOUR CONSECUTIVE gate columns of ONE batch row -- in the forward's unit-major column order exactly the
//     (i, f, g, o) of one unit, so the K-split partial products cross the workgroup as float4 records, and in the BPTT four
//     consecutive hidden units of one row, i.e. two ready-made partial-sum granules.
//   * forward (K-split): wave w gathers K-quarter w of h_{t-1} (rows x 64 granules of four units), 8 fragment
//     reads + 64 MFMAs, 8 float4 LDS writes, ONE barrier, every (row, unit) thread adds the four quarters and runs the cell.
//   * BPTT (reduce-scatter): a workgroup receives 32 senders x rows x 8 granules (four partial sums each) of partial dh
//     for its 32 units, sums them in registers + a transposing DPP / lane-swap butterfly, runs the gate-gradient math, publishes its dG image through LDS
//     (ONE barrier), multiplies it with its 128 gate rows of W_hh (4 fragment reads + 64 MFMAs) and sends the partial sums.
//   rows per group R <= 16; instantiated for RP = 4 / 8 / 16 (polls per lane, pairs per thread and the I/O block length follow).
//   * the activations the forward saves for the BPTT (gates i, f, g, o and the cell state) live in a WORKGROUP-MAJOR buffer:
//     saved[group][member][t]{ gates [R][32 units][4], c [R][32 units] } -- 640 R bytes per workgroup and timestep, consecutive
//     timesteps adjacent.  In the canonical [t][b][...] order every timestep of every array is its own page at B = 128 (2 MB apart),
//     and the address translations of a block's loads serialise IN FRONT of whatever the wave issues next: 3.8 us per 2-step
//     block in the 16-row BPTT (trace: profiles/r03k_*), 1.5 us per timestep of 7.9 (what-if build reading the same pages).
// LDS row pitches are = 2 (mod 16) 16-byte slots: ds_read_b128 serves lanes in groups {0-3,12-15,20-27}, ..., i.e. the 16 rows
// of one 8-k chunk and a neighbouring chunk, which a pitch of 2 slots spreads over all 16 slot classes.
#include "lv_device.h"
#include "lv_persist_common.h"

#if defined(LV_TRACE) && !defined(LV_EMU)
__device__ unsigned long long* lv_trace_buf = nullptr;
extern "C" int lv_trace_set(void* p) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(lv_trace_buf), &p, sizeof(p)); }
#endif

#ifndef LV_SBB16
#define LV_SBB16 8                      // (measurement knobs of profiles/microbench: input block length of the 16-row BPTT,
#endif                                  //  I/O block length and granules per lane and polling round of the 16-row forward)
#ifndef LV_HB16
#define LV_HB16 2                       // slot batches of the BPTT receive polled together (2: two rounds at 16 rows; 4: one)
#endif
#ifndef LV_SBK16
#define LV_SBK16 4
#endif
#ifndef LV_GJ16
#define LV_GJ16 16
#endif
#ifndef LV_FWD_Q
#define LV_FWD_Q 2                      // forward: h travels as 16-byte granules (four units, each dword = binary16 / bf16 bits under a 16-bit
#endif                                  // tag of its own): 2 = every instantiation, 1 = the 4-row one only, 0: 8-byte granules (A/B builds)
#ifndef LV_RS_Q
#define LV_RS_Q 2                       // BPTT: the reduce-scatter's partial sums travel as 16-byte granules (rs4_*: four 30-bit floats, 2 tag
#endif                                  // bits each): 2 = every instantiation, 1 = the 4-row one only, 0: 8-byte granules (A/B builds)
#ifndef LV_P16_ABL
#define LV_P16_ABL 0                    // measurement knob (profiles/microbench/lstm_anatomy_probe.py): WHAT-IF builds of the final
#endif                                  // kernels with one phase of a timestep removed -- results are garbage, the time is the point.
                                        // bit 0: no MFMAs; bit 1: hand-off tags not tested (no step waits for its producers: the
                                        // local pipeline alone); bit 2: no transcendental cell / gate-gradient math; bit 3: no
                                        // LDS quarter-sum exchange + barrier (forward) / dG image barrier (BPTT); bit 4 (forward): the
                                        // gathered granules are not staged through LDS (fragments straight from the poll registers);
                                        // bit 5 (forward): the granules are not loaded at all (bits 4, 5: builds with LV_FWD_Q = 0 only)

namespace {

using namespace lvp;

__device__ __forceinline__ float abl_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }      // (what-if builds only)

constexpr int HP16 = PH / 2 + 8;        // dwords per row of the gathered h image (130 slots: = 2 mod 16)
constexpr int DP16 = 64 + 8;            // dwords per row of the dG image (18 slots)
constexpr int RED_SLOTS = 33;           // float4 slots per row of a quarter product (32 units + 1: 8 consecutive rows = 8 slot classes)
constexpr int SAVED_PER_ROW = 160;     // floats per batch row in a (workgroup, timestep) record of the saved activations: 32 units x (4 gates + c)
constexpr int RS16_SLOTS_MAX = 256;     // granule slots of one (receiver, sender) pair at 16 rows (16 RP in general: the pairs are dense)
template <int V> struct lv_const { static constexpr int value = V; };

// ---- weight images ------------------------------------------------------------------------------------------------------------
// forward:  Wk16[wave_id (128) = 4m + w][ks (8)][nb (8)][lane (64)] uint4.  Lane (c = l & 15, kq = l >> 4) holds, for gate column
//           16 nb + c of workgroup m (unit 32m + ((16 nb + c) >> 2), gate c & 3), the 8 weights of k = 256w + 32ks + 8kq + e.
__global__ __launch_bounds__(256) void pack_w_k16_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk, int f16);
// BPTT:     Wrs16[wave_id (128) = 4m + w][ks (4)][nb (16)][lane (64)] uint4.  Lane (c, kq) holds, for output unit
//           j = 256w + 16 nb + c, the 8 weights W_hh[gate * H + unit][j] of the workgroup's local gate rows n'' = 32ks + 8kq + e
//           (unit = 32m + (n'' >> 2), gate = n'' & 3).
__global__ __launch_bounds__(256) void pack_w_rs16_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk);

// both images in one launch (the encoder's W_hh changes every inner step: blocks [0, n) pack the forward image, [n, 2n) the BPTT one)
// (f16: the image holds IEEE binary16 -- the forward kernel's F16 instantiation; 11 bits of significand instead of 8)
__device__ __forceinline__ void pack_w_k16_one(const float* __restrict__ whh, uint4* __restrict__ wpk, long idx, int f16 = 0) {
    const int l = (int)(idx & 63), nb = (int)((idx >> 6) & 7), ks = (int)((idx >> 9) & 7);
    const int wave_id = (int)(idx >> 12), m = wave_id >> 2, w = wave_id & 3;
    const int c = l & 15, kq = l >> 4, col = 16 * nb + c;
    const float* row = whh + ((long)(col & 3) * PH + 32 * m + (col >> 2)) * PH + 256 * w + 32 * ks + 8 * kq;
    if (f16) {
        float r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) r[e] = lv_sat_f16(row[e]);      // binary16 saturates, never inf; NaN stays NaN
        wpk[idx] = make_uint4(lv_pack_f16x2(r[0], r[1]), lv_pack_f16x2(r[2], r[3]), lv_pack_f16x2(r[4], r[5]), lv_pack_f16x2(r[6], r[7]));
    }
    else
        wpk[idx] = make_uint4(lv_pack_bf16x2(row[0], row[1]), lv_pack_bf16x2(row[2], row[3]), lv_pack_bf16x2(row[4], row[5]),
                              lv_pack_bf16x2(row[6], row[7]));
}
__device__ __forceinline__ void pack_w_rs16_one(const float* __restrict__ whh, uint4* __restrict__ wpk, long idx) {
    const int l = (int)(idx & 63), nb = (int)((idx >> 6) & 15), ks = (int)((idx >> 10) & 3);
    const int wave_id = (int)(idx >> 12), m = wave_id >> 2, w = wave_id & 3;
    const int c = l & 15, kq = l >> 4, j = 256 * w + 16 * nb + c;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int n2 = 32 * ks + 8 * kq + e;
        v[e] = whh[((long)(n2 & 3) * PH + 32 * m + (n2 >> 2)) * PH + j];
    }
    wpk[idx] = make_uint4(lv_pack_bf16x2(v[0], v[1]), lv_pack_bf16x2(v[2], v[3]), lv_pack_bf16x2(v[4], v[5]), lv_pack_bf16x2(v[6], v[7]));
}
__global__ __launch_bounds__(256) void pack_w_both16_kernel(const float* __restrict__ whh, uint4* __restrict__ wfwd, uint4* __restrict__ wbwd,
                                                            int fwd_f16) {
    const long n = 128L * 64 * 64;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) pack_w_k16_one(whh, wfwd, idx, fwd_f16);
    else if (idx < 2 * n) pack_w_rs16_one(whh, wbwd, idx - n);
}

__global__ __launch_bounds__(256) void pack_w_k16_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk, int f16) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < 128L * 8 * 8 * 64) pack_w_k16_one(whh, wpk, idx, f16);
}
__global__ __launch_bounds__(256) void pack_w_rs16_kernel(const float* __restrict__ whh, uint4* __restrict__ wpk) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx < 128L * 4 * 16 * 64) pack_w_rs16_one(whh, wpk, idx);
}

struct Fwd16P {
    const float* gx; const uint4* wpk;
    float* hs; float* cs; float* saved;
    gran_t* hx;                 // exchange: [2 parity][8 groups][16 rows][H/2] granules, zeroed before the launch
    int* status;
    int T, B, R;
    uint4* clr; long clr_n;     // double-buffered exchange (flags bit 1): the OTHER half, zeroed by this launch for the next one (clr_n uint4s)
};

// Zero `n` uint4s, the whole grid sharing the work (the first thing a persistent launch does, before a group without rows leaves):
// the exchange half the NEXT launch of this kind will poll.  Nobody reads it during this launch; the kernel boundary publishes it.
__device__ __forceinline__ void clear_other_half(uint4* q, long n) {
    if (!q) return;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
}

// =====================================================================================================================
// forward.  RP: rows per group this instantiation carries (4 / 8 / 16); NP = pairs (row, unit) per thread; SBK = timesteps per
// I/O block; GJ = granules per lane in flight per polling round.
template <int RP> struct Cfg16 {
    static constexpr int NP = RP > 8 ? 2 : 1;
    static constexpr int SBK = RP > 8 ? LV_SBK16 : 8;
    static constexpr int SBB = RP > 8 ? LV_SBB16 : 8; // BPTT: timesteps per input block
    static constexpr int GJ = RP > 8 ? LV_GJ16 : (RP > 4 ? 16 : 8);
};

template <int RP>
struct __attribute__((aligned(16))) Fwd16Lds {
    uint32_t hl[RP * HP16];                   // gathered h_{t-1}: [row][k/2]; wave w owns dwords [128w, 128w + 128) of every row
    f32x4 red[2][4][RP][RED_SLOTS];           // [step parity][wave]: quarter product, (i, f, g, o) of [row][unit of the workgroup]
    uint32_t dump[256];                       // where the lanes of a ragged gather round put what they did not need (no branch per granule)
    int abort;
};

template <int RP, bool LOCAL, bool F16 = false>
__global__ __launch_bounds__(256) void lstm_fwd_persist_k16_kernel(Fwd16P p) {
    // LOCAL: hand-off stores without the agent-scope write-through (lv_xcd_store_u64): valid while a group's 32 workgroups share an XCD
    // F16: the recurrent operands -- the register image of W_hh and the h granules -- are IEEE binary16 and the products run on
    // v_mfma_f32_16x16x32_f16: the same kernel at 11 instead of 8 bits of significand for the WEIGHTS, whose rounding acts at every
    // timestep alike (the encoder's forward: 1.4e-4 -> 2e-5 relative on the KL, profiles/r05a_kl_ablation.txt); h in (-1, 1) and
    // LSTM weights are far inside binary16's range.  The BPTT keeps bf16 (gradients need the exponent range).
    auto h16 = [](float v) -> uint32_t { return F16 ? (uint32_t)lv_f32_to_f16_bits(v) : lv_f32_to_bf16_bits(v); };
    auto put = [](gran_t* q, gran_t v) { if (LOCAL) lv_xcd_store_u64(q, v); else gran_store(q, v); };
    constexpr int NP = Cfg16<RP>::NP, SBK = Cfg16<RP>::SBK, GJ = Cfg16<RP>::GJ;
    LV_BLOCK_SHARED(Fwd16Lds<RP>, sm);
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);
    clear_other_half(p.clr, p.clr_n);
    if (rows == 0) return;                                    // a group without rows leaves its XCD to whoever else wants it
    if (tid == 0) s_abort = 0;

    uint4 wreg[8][8];                                         // [ks][nb]: 256 VGPRs, resident for the whole call
    {
        const uint4* wp = p.wpk + (long)wave_id * 8 * 8 * 64 + l;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) wreg[ks][nb] = wp[(ks * 8 + nb) * 64];
    }

    // this thread's (row, unit) pairs: rows tid >> 5 (+ 8 for the second pair), unit tid & 31 of the workgroup
    const int uw = tid & 31, punit = 32 * member + uw;
    const long BH = (long)B * PH;
    const long rec = (long)R * SAVED_PER_ROW;                  // floats of one (workgroup, timestep) record of the saved activations
    float* const sv = p.saved + (long)(group * PMEMBERS + member) * T * rec;
    int prow[NP]; bool own[NP]; long pidx[NP]; float c_state[NP]; int sg[NP], sc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        prow[q] = (tid >> 5) + 8 * q;
        own[q] = prow[q] < rows;
        pidx[q] = (long)(b0 + (own[q] ? prow[q] : 0)) * PH + punit;
        c_state[q] = own[q] ? p.cs[pidx[q]] : 0.f;
        sg[q] = (prow[q] * 32 + uw) * 4;                       // gates record of the pair, and its cell state behind the R rows of gates
        sc[q] = R * 128 + prow[q] * 32 + uw;
    }
    gran_t* const hx_g = p.hx + (long)group * 16 * (PH / 2);
    const long hx_par = (long)PGROUPS * 16 * (PH / 2);
    const bool even = !(uw & 1);

    // QF: a granule is 16 bytes = the four units 4i .. 4i + 3 of a row, every dword its unit's 16 bits under the low 16 bits of
    // the state's tag -- a granule is valid when all four dwords carry it, so 16-byte atomicity is assumed nowhere.  Half as many load
    // instructions per poll for the same bytes; the publishing lane of a quad collects its neighbours' dwords by DPP.
    constexpr bool QF = LV_FWD_Q == 2 || (RP == 4 && LV_FWD_Q == 1);
    auto tagq = [](int state) -> uint32_t { return 1u + (uint32_t)state % 0xFFFFu; };
    auto publish_q = [&](int state, int q, float hval) {
        const uint32_t dw = h16(hval) | (tagq(state) << 16);
        const uint32_t d1 = lv_quad_bcast_u32<1>(dw), d2 = lv_quad_bcast_u32<2>(dw), d3 = lv_quad_bcast_u32<3>(dw);
        if (own[q] && !(uw & 3)) {
            char* d = reinterpret_cast<char*>(hx_g + (long)(state & 1) * hx_par + (long)prow[q] * (PH / 2)) + (punit >> 2) * 16;
            if (LOCAL) lv_xcd_store_q4(d, make_uint4(dw, d1, d2, d3)); else lv_agent_store_q4(d, make_uint4(dw, d1, d2, d3));
        }
    };
#pragma unroll
    for (int q = 0; q < NP; ++q) {      // publish the initial state hs[0] as state 0 (tag 1)
        if constexpr (QF) { publish_q(0, q, own[q] ? p.hs[pidx[q]] : 0.f); continue; }
        const uint32_t mine = h16(own[q] ? p.hs[pidx[q]] : 0.f);
        const uint32_t next = lv_lane_xor1_u32(mine);      // (even lanes: their right-hand neighbour; DPP, no LDS round trip)
        if (own[q] && even) put(hx_g + (long)prow[q] * (PH / 2) + (punit >> 1), ((gran_t)1u << 32) | (gran_t)(mine | (next << 16)));
    }

    float4 gxb[NP][SBK];
    f32x4 recb[NP][SBK];
    float cb[NP][SBK], hb[NP][SBK];
    auto load_slots = [&](int tb, int lo, int hi) {          // gx of steps tb + [lo, hi) into their slots
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s2 = 0; s2 < SBK; ++s2) {
                if (s2 < lo || s2 >= hi) continue;
                // UNCONDITIONAL, from a clamped address (steps >= T never run, pairs nobody owns read row 0 of the slice and are
                // never used): behind an `if` the loaded value meets a zero in a phi, the four components stop being one register
                // tuple, and the compiler copies them out right behind the load -- s_waitcnt vmcnt(0) after EVERY block load, a
                // full HBM round trip each (16-row BPTT: 3.8 us per 2-step block).
                const int t = tb + s2 < T ? tb + s2 : T - 1;
                gxb[q][s2] = *reinterpret_cast<const float4*>(p.gx + ((long)t * BH + pidx[q]) * 4);
            }
    };
    auto store_block = [&](int tb) {
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int s2 = 0; s2 < SBK; ++s2) {
                const int t = tb + s2;
                if (own[q] && t < T) {
                    lv_store_nt(recb[q][s2],
                                reinterpret_cast<f32x4*>(sv + (long)t * rec + sg[q]));
                    lv_store_nt(cb[q][s2], sv + (long)t * rec + sc[q]);
                    lv_store_nt(hb[q][s2], p.hs + (long)(t + 1) * BH + pidx[q]);
                    if (t == T - 1) p.cs[(long)T * BH + pidx[q]] = cb[q][s2];      // the final cell state in the canonical place
                }
            }
    };
    load_slots(0, 0, SBK);
    __syncthreads();

    const int nq = rows * 128;                                 // granules of this wave's K quarter
    const int brow = (l & 15) < rows ? (l & 15) : 0;           // batch row of this lane's B fragments (rows beyond the slice re-read row 0)
    const int kq = l >> 4;
    uint32_t abl_keep[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};      // (LV_P16_ABL bit 4 only)
    for (int tb = 0; tb < T; tb += SBK) {
#pragma unroll
        for (int s2 = 0; s2 < SBK; ++s2) {
            const int t = tb + s2;
            if (t >= T) break;
            LV_TRACE_MARK(t, 0);
            // ---- gather K-quarter w of state t (tag t + 1) into this wave's part of the LDS image ---------------------------
            const gran_t* src = hx_g + (long)(t & 1) * hx_par + 128 * w;
            const uint32_t want = (uint32_t)(t + 1);
            if constexpr (QF) {
                // lane l takes granule 64w + l (k = 256w + 4l .. + 3) of each of the slice's rows -- every load instruction reads 1 KB
                // of one row -- in polling rounds of 4 (4-row instantiation) or 8 rows; rows the slice does not have are neither tested
                // nor staged
                constexpr int QR = RP == 4 ? 4 : 8;
                const char* sq = reinterpret_cast<const char*>(hx_g + (long)(t & 1) * hx_par) + (64 * w + l) * 16;
                const long rowb = (long)(PH / 2) * 8;
                const uint32_t wq = tagq(t) << 16;
                auto round = [&](auto R0) {      // (a lambda per round: a loop over rounds around the polling loop is not unrolled)
                    constexpr int r0 = decltype(R0)::value;
                    if (r0 >= rows) return;
                    uint4 v[QR];
                    int spins = 0;
                    bool ok;
                    do {
                        if constexpr (QR == 4)
                            lv_agent_load_q4x4(sq, sq + (rows > 1 ? rowb : 0), sq + (rows > 2 ? 2 * rowb : 0), sq + (rows > 3 ? 3 * rowb : 0), v);
                        else {
                            // (rows the slice does not have are loaded from their own -- allocated, unused -- rows of the exchange)
                            static_assert((PH / 2) * 8 == 4096, "lv_agent_load_q4x8_rows: rows 4 KB apart");
                            lv_agent_load_q4x8_rows(sq + r0 * rowb, reinterpret_cast<uint4 (&)[8]>(v));
                        }
                        uint32_t x = 0u;
#pragma unroll
                        for (int j = 0; j < QR; ++j)
                            if (r0 + j < rows) x |= (v[j].x ^ wq) | (v[j].y ^ wq) | (v[j].z ^ wq) | (v[j].w ^ wq);
                        ok = (LV_P16_ABL & 2) ? true : (x >> 16) == 0u;
                        ok = __all(ok);
                        if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; break; }
                        if (!ok) lv_poll_backoff();
                    } while (!ok);
                    LV_TRACE_VAL(t, 6, spins);
#pragma unroll
                    for (int j = 0; j < QR; ++j)
                        if (r0 + j < rows)
                            *reinterpret_cast<uint2*>(&sm.hl[(r0 + j) * HP16 + 128 * w + 2 * l]) =
                                make_uint2((v[j].x & 0xFFFFu) | (v[j].y << 16), (v[j].z & 0xFFFFu) | (v[j].w << 16));
                };
                round(lv_const<0>());
                if constexpr (RP > QR) round(lv_const<QR>());
            } else
            for (int base = 0; base < nq; base += 64 * GJ) {
                // every poll round issues ALL its loads before it looks at a tag (first build: a load and its tag test per granule
                // inside one predicated block compiled to load -> wait -> compare, GJ dependent round trips: 6.8 us per step at 8 rows)
                gran_t v[GJ];
                int spins = 0;
                bool ok;
                do {
                    ok = true;
#pragma unroll
                    for (int j = 0; j < GJ; ++j) {
                        const int q = base + j * 64 + l;
                        const bool in = q < nq;
                        if constexpr (LV_P16_ABL & 32) { v[j] = ((gran_t)want << 32) | (gran_t)(uint32_t)(q + t); continue; }
                        v[j] = gran_load(src + (in ? (q >> 7) * (PH / 2) + (q & 127) : 0));      // out-of-range lanes re-read granule 0: no branch
                        if (!(LV_P16_ABL & 2)) ok = ok && (!in || (uint32_t)(v[j] >> 32) == want);
                    }
                    ok = __all(ok);
                    if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; break; }
                } while (!ok);
                LV_TRACE_VAL(t, 6, spins);
                if constexpr (LV_P16_ABL & 16) {
#pragma unroll
                    for (int j = 0; j < GJ; ++j) abl_keep[j & 7] ^= (uint32_t)v[j];
                } else {
#pragma unroll
                for (int j = 0; j < GJ; ++j) {
                    const int q = base + j * 64 + l;
                    uint32_t* dstw = q < nq ? &sm.hl[(q >> 7) * HP16 + 128 * w + (q & 127)] : &sm.dump[tid];
                    *dstw = (uint32_t)v[j];
                }
                }
            }
            // Bulk I/O goes out right BEHIND a completed gather: a wave's loads and stores retire in order, so whatever is issued in
            // front of a poll is waited for by that poll with its full memory latency (first build: everything at the block
            // boundary).  First step of a block: the previous block's results and the one gx slot that was still in use; last
            // step: the next block's other slots.
            if (s2 == 0 && tb > 0) { store_block(tb - SBK); load_slots(tb, SBK - 1, SBK); }
            if (s2 == SBK - 1) load_slots(tb + SBK, 0, SBK - 1);
            LV_TRACE_MARK(t, 1);
            LV_WAIT_LDS();                                     // the wave reads back only what its own lanes wrote
            LV_TRACE_MARK(t, 2);

            // ---- this wave's K quarter of the product: A = weights (16 gate columns), B = h (16 batch rows) ------------------
            f32x4 acc[8];
            const uint4* bp = reinterpret_cast<const uint4*>(sm.hl + brow * HP16 + 128 * w) + kq;
            uint4 bfr[8];
            if constexpr (LV_P16_ABL & 16) {
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) bfr[ks] = make_uint4(abl_keep[ks], abl_keep[(ks + 1) & 7], abl_keep[(ks + 2) & 7], abl_keep[(ks + 3) & 7]);
            } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bfr[ks] = bp[ks * 4];
            }
            LV_SCHED_BARRIER();                                // all eight fragment reads in flight before the first MFMA (left alone the
                                                               // compiler reads each one right before its eight MFMAs, behind lgkmcnt(0))
            if constexpr (LV_P16_ABL & 1) {
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) acc[nb] = f32x4{abl_u2f(bfr[nb].x), abl_u2f(bfr[nb].y), 0.f, 0.f};
            } else {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks)
#pragma unroll
                for (int nb = 0; nb < 8; ++nb)
                    if constexpr (F16)
                        acc[nb] = ks == 0 ? lv_mfma_16x16x32_f16_areg_first(wreg[0][nb], bfr[0]) : lv_mfma_16x16x32_f16_areg(wreg[ks][nb], bfr[ks], acc[nb]);
                    else
                        acc[nb] = ks == 0 ? lv_mfma_16x16x32_bf16_areg_first(wreg[0][nb], bfr[0]) : lv_mfma_16x16x32_bf16_areg(wreg[ks][nb], bfr[ks], acc[nb]);
            LV_MFMA_DRAIN();
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) LV_MFMA_RESULT(acc[nb]);
            }
            // D: lane (c = l & 15: batch row, rq = l >> 4) holds gate columns 16 nb + 4 rq + r = the (i, f, g, o) of unit 4 nb + rq
            if ((l & 15) < RP) {
                f32x4* rd = sm.red[t & 1][w][l & 15];
#pragma unroll
                for (int nb = 0; nb < 8; ++nb) rd[4 * nb + kq] = acc[nb];
            }
            LV_TRACE_MARK(t, 3);
            if (!(LV_P16_ABL & 8))
            __syncthreads();                                   // the four quarter products (double-buffered by step parity); compiles to
                                                               // lgkmcnt(0) + s_barrier: the block loads issued above stay in flight
            LV_TRACE_MARK(t, 4);
            if (s_abort) { if (tid == 0) atomicExch(p.status, 100 + t); return; }

            // ---- cell update and hand-off of h_t ----------------------------------------------------------------------------------
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                float h = 0.f;
                if (own[q]) {
                    const f32x4 q0 = sm.red[t & 1][0][prow[q]][uw], q1 = sm.red[t & 1][1][prow[q]][uw];
                    const f32x4 q2 = sm.red[t & 1][2][prow[q]][uw], q3 = sm.red[t & 1][3][prow[q]][uw];
                    const float4 gxv = gxb[q][s2];
                    auto sg_ = [](float x) { return (LV_P16_ABL & 4) ? 0.5f + 0.25f * x : lv_sigmoid_fast(x); };
                    auto th_ = [](float x) { return (LV_P16_ABL & 4) ? 0.9f * x : lv_tanh_fast(x); };
                    const float ig = sg_(gxv.x + ((q0[0] + q1[0]) + (q2[0] + q3[0])));
                    const float fg = sg_(gxv.y + ((q0[1] + q1[1]) + (q2[1] + q3[1])));
                    const float gg = th_(gxv.z + ((q0[2] + q1[2]) + (q2[2] + q3[2])));
                    const float og = sg_(gxv.w + ((q0[3] + q1[3]) + (q2[3] + q3[3])));
                    const float c = fg * c_state[q] + ig * gg;
                    h = og * th_(c);
                    c_state[q] = c;
                    recb[q][s2] = f32x4{ig, fg, gg, og};
                    cb[q][s2] = c; hb[q][s2] = h;
                }
                if constexpr (QF) { publish_q(t + 1, q, h); continue; }
                const uint32_t mine = h16(h);
                const uint32_t next = lv_lane_xor1_u32(mine);      // (even lanes: their right-hand neighbour; DPP, no LDS round trip)
                if (own[q] && even)
                    put(hx_g + (long)((t + 1) & 1) * hx_par + (long)prow[q] * (PH / 2) + (punit >> 1),
                        ((gran_t)(uint32_t)(t + 2) << 32) | (gran_t)(mine | (next << 16)));
            }
            LV_TRACE_MARK(t, 5);
        }
    }
    store_block((T - 1) / SBK * SBK);
}

// =====================================================================================================================
// BPTT, reduce-scatter hand-off.  Exchange buffer: [parity][group][receiver (32)][sender (32)][16 RP granules], DENSE for the
// instantiation (a first layout with a fixed 2 KB stride per pair used 512 B of every 2 KB at 4 rows and 7 us per timestep: a
// quarter of the L2 channels carried all the traffic).  A sender lane holds, for batch row c and column block nb, the partial dh of
// units u = 16 b + 4 rq + 2 h2 + {0, 1} of receiver 8w + (nb >> 1) (b = nb & 1; rq = l >> 4; h2 = the register pair): slot
// s = (2 b + h2) 4 RP + 4 c + rq, so that ONE store instruction (fixed b, h2) writes one contiguous run of 32 RP bytes -- whole
// 128-byte lines, not four partial writes per line.  A receiver wave w sums the slots [4 RP w, 4 RP w + 4 RP) (b = w >> 1,
// h2 = w & 1) in batches of 16 (batch beta: rows 4 beta + (p >> 2), rq = p & 3 for lane position p = l & 15): lane group l >> 4 =
// eight of the 32 senders, two shuffles add the four groups.  Owners: lanes 0..31 take batch 2q, lanes 32..63 batch 2q + 1
// (q = pair index), low / high half of the granule by (l >> 4) & 1.
// (That is the 8-BYTE granule form, LV_RS_Q = 0, kept for A/B builds.  The shipped form moves the same bytes of the same dense
//  per-pair extent as 16-byte granules of four units each -- layout, lane maps and the transposing sum: the QB comments in the kernel.)
struct Bwd16P {
    const float* dh_ext; const float* dh_last;
    const uint4* wpk;
    const float* saved; const float* cs; const float* hs;
    uint16_t* dG16; float* dGsum;
    float* dh0; float* dc0; int tanh_init;
    gran_t* gxch;
    int* status;
    int T, B, R;
    uint4* clr; long clr_n;     // as Fwd16P
};

template <int RP>
struct __attribute__((aligned(16))) Bwd16Lds {
    uint32_t dgl[2][16 * DP16];                         // [step parity] dG of this workgroup's 128 gate rows (B operand image), rows < R valid
    uint16_t og[2][RP][4][32];                          // [step parity] the same dG as it goes to memory: [row][gate][unit in WG]
    int abort;
};

template <int RP, bool LOCAL>
__global__ __launch_bounds__(256) void lstm_bwd_persist_rs16_kernel(Bwd16P p) {
    auto put = [](gran_t* q, gran_t v) { if (LOCAL) lv_xcd_store_u64(q, v); else gran_store(q, v); };
    constexpr int NP = Cfg16<RP>::NP, SBK = Cfg16<RP>::SBB;
    constexpr int NB = RP / 4;                          // slot batches of 16 a wave receives
    constexpr int SLOTS = 16 * RP;                      // granules of one (receiver, sender) pair
    LV_BLOCK_SHARED(Bwd16Lds<RP>, sm);
    int& s_abort = sm.abort;
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int group = (int)blockIdx.x % PGROUPS, member = (int)blockIdx.x / PGROUPS;
    const int wave_id = member * 4 + w;
    const int B = p.B, R = p.R, T = p.T;
    const int b0 = group * R;
    const int rows = (b0 >= B) ? 0 : ((B - b0) < R ? (B - b0) : R);
    clear_other_half(p.clr, p.clr_n);
    if (rows == 0) return;
    if (tid == 0) s_abort = 0;
    for (int i = tid; i < 2 * 16 * DP16; i += 256) sm.dgl[0][i] = 0u;      // rows the slice does not have multiply as zeros

    uint4 wreg[4][16];                                  // [ks][nb]
    {
        const uint4* wp = p.wpk + (long)wave_id * 4 * 16 * 64 + l;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int nb = 0; nb < 16; ++nb) wreg[ks][nb] = wp[(ks * 16 + nb) * 64];
    }

    // owner pairs of this lane: batch beta = 2q + (l >> 5) -> batch row 4 beta + (p >> 2); unit from (w, p, granule half)
    // (QB, 16-byte granules: a pair's 128 RP bytes are 8 RP granules, granule (unit block b, row c, unit quad rq) at (b RP + c) 4 + rq;
    //  wave w receives the 2 RP granules from 2 RP w on in ROUNDS of eight (lane l: granule l & 7 of the round, senders
    //  4 (l >> 3) + 0..3), and the transposing butterfly leaves the lane with unit 2 ((l >> 4) & 1) + (l >> 5) of its quad l & 3.
    //  4 rows: one round, the lanes with bit 3 clear own (row 2 (w & 1) + ((l >> 2) & 1)); 8 / 16 rows: pair q of a lane comes from
    //  round 2q + ((l >> 3) & 1), row (RP / 2)(w & 1) + 4q + 2 ((l >> 3) & 1) + ((l >> 2) & 1), and every lane owns.)
    constexpr bool QB = LV_RS_Q == 2 || (RP == 4 && LV_RS_Q == 1);
    constexpr bool Q4 = RP == 4 && QB;
    const int pp = l & 15;
    const int uw = QB ? 16 * (w >> 1) + 4 * (l & 3) + 2 * ((l >> 4) & 1) + (l >> 5)
                      : 16 * (w >> 1) + 4 * (pp & 3) + 2 * (w & 1) + ((l >> 4) & 1);
    const int punit = 32 * member + uw;
    const long BH = (long)B * PH;
    const long rec = (long)R * SAVED_PER_ROW;
    const float* const sv = p.saved + (long)(group * PMEMBERS + member) * T * rec;
    int prow[NP]; bool own[NP]; long pidx[NP]; int sg[NP], sc[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int beta = 2 * q + (l >> 5);
        prow[q] = Q4 ? 2 * (w & 1) + ((l & 7) >> 2) : QB ? (RP / 2) * (w & 1) + 4 * q + 2 * ((l >> 3) & 1) + ((l >> 2) & 1) : 4 * beta + (pp >> 2);
        own[q] = (Q4 ? !(l & 8) : QB ? true : beta < NB) && prow[q] < rows;
        pidx[q] = (long)(b0 + (own[q] ? prow[q] : 0)) * PH + punit;
        sg[q] = ((own[q] ? prow[q] : 0) * 32 + uw) * 4;
        sc[q] = R * 128 + (own[q] ? prow[q] : 0) * 32 + uw;
    }
    const long px_par = (long)PGROUPS * PMEMBERS * PMEMBERS * SLOTS;
    gran_t* const px_g = p.gxch + (long)group * PMEMBERS * PMEMBERS * SLOTS;
    // receive: sender 8 (l >> 4) + j, slot 4 RP w + 16 beta + p
    const gran_t* const rx = px_g + ((long)member * PMEMBERS + 8 * (l >> 4)) * SLOTS + 4 * RP * w + pp;
    // send: lane (c = l & 15: batch row, rq = l >> 4), column block nb, register pair h2 -> receiver 8w + (nb >> 1),
    // slot (2 (nb & 1) + h2) 4 RP + 4 c + rq
    gran_t* const tx = px_g + ((long)(8 * w) * PMEMBERS + member) * SLOTS + 4 * (l & 15) + (l >> 4);
    // RP = 4, merged stores: lane position (l & 15) = c + 4 j carries batch row c of column block 4 n4 + j -> receiver
    // 8w + 2 n4 + (j >> 1), slot (2 (j & 1) + h2) 16 + 4 c + rq
    gran_t* const tx4 = px_g + ((long)(8 * w + (((l & 15) >> 2) >> 1)) * PMEMBERS + member) * SLOTS + 2 * (((l & 15) >> 2) & 1) * 4 * RP +
                        4 * (l & 3) + (l >> 4);

    // QB receive: senders 4 (l >> 3) + j, granule 2 RP w + 8 round + (l & 7).  Send, 4 rows: lane position (l & 15) = c + 4 j of a
    // merged chunk n4 carries the quad (row c, rq = l >> 4) of column block 4 n4 + j -> receiver 8w + 2 n4 + (j >> 1), unit block
    // j & 1; 8 / 16 rows: lane (c = l & 15, rq = l >> 4) sends the quad of column block nb to receiver 8w + (nb >> 1), unit block nb & 1.
    const char* const rx_q = reinterpret_cast<const char*>(px_g) + ((long)member * PMEMBERS + 4 * (l >> 3)) * SLOTS * 8 + (2 * RP * w + (l & 7)) * 16;
    char* const tx_q = RP == 4 ? reinterpret_cast<char*>(px_g) + ((long)(8 * w + (((l & 15) >> 2) >> 1)) * PMEMBERS + member) * SLOTS * 8 +
                                     ((((l & 15) >> 2) & 1) * 16 + 4 * (l & 3) + (l >> 4)) * 16
                               : reinterpret_cast<char*>(px_g) + ((long)(8 * w) * PMEMBERS + member) * SLOTS * 8 + (4 * (l & 15) + (l >> 4)) * 16;
    float dc_rec[NP], gsum[NP][4];
    float dhb[NP][SBK], ctb[NP][SBK + 1];
    float4 recb[NP][SBK];
#pragma unroll
    for (int q = 0; q < NP; ++q) { dc_rec[q] = 0.f; gsum[q][0] = gsum[q][1] = gsum[q][2] = gsum[q][3] = 0.f; }
    const bool has_ext = p.dh_ext != nullptr;
    const float* const dh_src = has_ext ? p.dh_ext : p.cs;      // (without dh_ext the loads still run, on any mapped [T][B][H]-sized array)
    auto load_block = [&](int t_hi) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
#pragma unroll
            for (int s2 = 0; s2 < SBK; ++s2) {
                const int t = t_hi - s2 < 0 ? 0 : t_hi - s2;      // unconditional loads from clamped addresses: see the forward's load_slots
                dhb[q][s2] = dh_src[(long)t * BH + pidx[q]];
                recb[q][s2] = *reinterpret_cast<const float4*>(sv + (long)t * rec + sg[q]);
            }
#pragma unroll
            for (int s2 = 0; s2 <= SBK; ++s2) {
                const int t = t_hi - s2;
                const float* src = t >= 0 ? sv + (long)t * rec + sc[q] : p.cs + pidx[q];      // c_t; c_{-1} = the initial state
                ctb[q][s2] = *src;
            }
        }
    };
    // dG16 of one step, after that step's barrier: 16-byte chunks [row][gate][quarter of the 32 units] of the og image
    auto store_step = [&](int par, int t) {
        for (int c = tid; c < RP * 16; c += 256) {
            const int qq = c & 3, g = (c >> 2) & 3, r = c >> 4;
            if (r < rows) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(&sm.og[par][r][g][8 * qq]);      // 8 bf16, moved as raw bits
                uint16_t* dst = p.dG16 + ((long)t * B + (b0 + r)) * 4 * PH + (long)g * PH + 32 * member + 8 * qq;
                lv_store_nt(v, reinterpret_cast<f32x4*>(dst));
            }
        }
    };
    load_block(T - 1);
    __syncthreads();

    const int brow = l & 15, kq = l >> 4;                // B-fragment row of this lane (rows beyond the slice are zero in the image)
    const bool closing = p.dh0 || p.tanh_init;

    // receive phase k: the 32 senders' partial sums for this wave's NB batch rows x 32 units; owners get their pairs' totals.
    // Two batch rows (16 granules per lane) are polled at a time: all 32 granules of the 16-row form in flight at once would
    // need 64 registers next to the 256 of the weights (the first build spilled 92); the sums over the senders are taken in a
    // fixed order once a round is complete, so the result does not depend on arrival order.
    constexpr int HB = NB < LV_HB16 ? NB : LV_HB16;
    LV_TRACE_ONLY(int tr_spins[2] = {0, 0};)
    auto receive_round = [&](auto H0, const gran_t* src, uint32_t want, float (&dh_rec)[NP]) -> bool {
        constexpr int h0 = decltype(H0)::value;
        gran_t v[HB][8];
        int spins = 0;
        bool ok;
        do {                                          // all loads of the round in flight, then the tags
            ok = true;
#pragma unroll
            for (int bt = 0; bt < HB; ++bt)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v[bt][j] = gran_load(src + (long)j * SLOTS + 16 * (h0 + bt));
                    if (!(LV_P16_ABL & 2)) ok = ok && (uint32_t)(v[bt][j] >> 56) == want;
                }
            ok = __all(ok);
            if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; return false; }
        } while (!ok);
        LV_TRACE_ONLY(tr_spins[h0 ? 1 : 0] = spins;)
#pragma unroll
        for (int bt = 0; bt < HB; ++bt) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a += rs_lo(v[bt][j]); b += rs_hi(v[bt][j]); }
            a = lv_add_xor16(a); b = lv_add_xor16(b);      // the four lane groups' (= 4 x 8 senders') sums, on the lane-swap instructions
            a = lv_add_xor32(a); b = lv_add_xor32(b);
            // batch h0 + bt belongs to pair q = (h0 + bt) >> 1 of the lanes with (l >> 5) == ((h0 + bt) & 1)
            if ((l >> 5) == ((h0 + bt) & 1)) dh_rec[(h0 + bt) >> 1] = (l & 16) ? b : a;
        }
        return true;
    };
    // Q4: ONE polling round of four 16-byte granules per lane (senders 4 (l >> 3) + 0..3); the four components are summed over the
    // senders in a fixed order -- in the lane, across the two lane groups of a row (DPP), then by two transposing lane-swap steps
    // that leave every lane with the total of ONE unit: 10 cross-lane instructions, none of them an LDS round trip.
    auto receive_q4 = [&](int k, float (&dh_rec)[NP]) -> bool {
        const char* src = rx_q + (long)(k & 1) * px_par * 8;
        const uint32_t want = rs4_tag(k);
        uint4 u[4];
        int spins = 0;
        bool ok;
        do {
            lv_agent_load_q4x4(src, src + (long)SLOTS * 8, src + (long)2 * SLOTS * 8, src + (long)3 * SLOTS * 8, u);
            ok = (LV_P16_ABL & 2) ? true : rs4_all_tagged(u, want);
            ok = __all(ok);
            if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; return false; }
            if (!ok) lv_poll_backoff();
        } while (!ok);
        LV_TRACE_ONLY(tr_spins[0] = spins;)
        float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { a += rs4_val(u[j].x); b += rs4_val(u[j].y); c += rs4_val(u[j].z); d += rs4_val(u[j].w); }
        a = lv_add_xor8(a); b = lv_add_xor8(b); c = lv_add_xor8(c); d = lv_add_xor8(d);
        dh_rec[0] = lv_fold32(lv_fold16(a, c), lv_fold16(b, d));      // lane (half h, row parity rp): component 2 rp + h
        return true;
    };
    // 8 / 16 rows: polls of TWO rounds (8 granules per lane in flight, one statement); the first butterfly step transposes too
    // (lv_fold8: the lanes with bit 3 clear keep the first round's sums, the others the second's), so every lane ends with a total
    // of its own -- pair q of the lane from the poll q.
    auto receive_q8 = [&](int k, float (&dh_rec)[NP]) -> bool {
        const char* src = rx_q + (long)(k & 1) * px_par * 8;
        const uint32_t want = rs4_tag(k);
        auto one = [&](auto Q) -> bool {
            constexpr int q = decltype(Q)::value;
            uint4 u[8];
            int spins = 0;
            bool ok;
            do {
                lv_agent_load_q4x8_rs<SLOTS * 8>(src + 256 * q, u);      // rounds 2q, 2q + 1: granules 16q + 8 (i >> 2) on, senders i & 3
                ok = (LV_P16_ABL & 2) ? true : rs4_all_tagged(u, want);
                ok = __all(ok);
                if (!ok && ++spins > SPIN_LIMIT) { s_abort = 1; return false; }
                if (!ok) lv_poll_backoff();
            } while (!ok);
            LV_TRACE_ONLY(tr_spins[q ? 1 : 0] = spins;)
            float a[2], b[2], c[2], d[2];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                a[r] = b[r] = c[r] = d[r] = 0.f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[r] += rs4_val(u[4 * r + j].x); b[r] += rs4_val(u[4 * r + j].y);
                    c[r] += rs4_val(u[4 * r + j].z); d[r] += rs4_val(u[4 * r + j].w);
                }
            }
            dh_rec[q] = lv_fold32(lv_fold16(lv_fold8(a[0], a[1]), lv_fold8(c[0], c[1])), lv_fold16(lv_fold8(b[0], b[1]), lv_fold8(d[0], d[1])));
            return true;
        };
        if (!one(lv_const<0>())) return false;
        if constexpr (NP > 1) { if (!one(lv_const<1>())) return false; }
        return true;
    };
    auto receive = [&](int k, float (&dh_rec)[NP]) -> bool {
        if constexpr (Q4) return receive_q4(k, dh_rec);
        else if constexpr (QB) return receive_q8(k, dh_rec);
        const gran_t* src = rx + (long)(k & 1) * px_par;
        const uint32_t want = rs_tag(k);
#pragma unroll
        for (int q = 0; q < NP; ++q) dh_rec[q] = 0.f;
        if (!receive_round(lv_const<0>(), src, want, dh_rec)) return false;
        if constexpr (NB > HB) {
            if (!receive_round(lv_const<HB>(), src, want, dh_rec)) return false;
        }
        if constexpr (NB > 2 * HB) {                  // (LV_HB16 = 1 at 16 rows: four rounds; the shipped HB = 2 stops above)
            if (!receive_round(lv_const<2 * HB>(), src, want, dh_rec)) return false;
        }
        if constexpr (NB > 3 * HB) {
            if (!receive_round(lv_const<3 * HB>(), src, want, dh_rec)) return false;
        }
        return true;
    };
    // multiply the dG image of step parity `par` with this wave's 256 output units and send phase k to their owners
    auto send = [&](int par, int k) {
        const uint4* bp = reinterpret_cast<const uint4*>(sm.dgl[par] + brow * DP16) + kq;
        uint4 bfr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) bfr[ks] = bp[ks * 4];
        LV_SCHED_BARRIER();
        const uint32_t tag = rs_tag(k);
        gran_t* dst = tx + (long)(k & 1) * px_par;
        gran_t* dst4 = tx4 + (long)(k & 1) * px_par;
        // Four column blocks at a time, software-pipelined by hand: the 16 MFMAs of chunk n4 + 1 are issued BEFORE the granules
        // of chunk n4 are merged, packed and stored (the weights are read from AGPRs by inline-assembly MFMAs, which the compiler
        // neither reorders nor guards: two accumulator sets, and one drain in front of the last chunk's reads).
        auto chunk_mfma = [&](int n4, f32x4 (&acc)[4]) {
            if constexpr (LV_P16_ABL & 1) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[j] = f32x4{abl_u2f(bfr[j].x + n4), abl_u2f(bfr[j].y), 0.f, 0.f};
                return;
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[j] = ks == 0 ? lv_mfma_16x16x32_bf16_areg_first(wreg[0][4 * n4 + j], bfr[0])
                                     : lv_mfma_16x16x32_bf16_areg(wreg[ks][4 * n4 + j], bfr[ks], acc[j]);
        };
        auto chunk_send = [&](int n4, const f32x4 (&acc)[4]) {
            if constexpr (RP == 4) {
                // Only lanes 0..3 of every 16-lane row hold batch rows at RP = 4: the four column blocks' quarter-rows are merged
                // into ONE fully populated register set (DPP row_shr into banks 1..3), so that this chunk goes out as 2 full-wave
                // stores instead of 8 quarter-wave ones -- the sends are bound by store INSTRUCTIONS, not bytes.
                float m[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    m[r] = acc[0][r];
                    m[r] = lv_row_shr_into<4, 1>(m[r], acc[1][r]);
                    m[r] = lv_row_shr_into<8, 2>(m[r], acc[2][r]);
                    m[r] = lv_row_shr_into<12, 3>(m[r], acc[3][r]);
                }
                if constexpr (Q4) {      // ... and as ONE 16-byte store: the lane's four consecutive units of its row are a granule
                    char* d = tx_q + (long)(k & 1) * px_par * 8 + (long)(2 * n4) * PMEMBERS * SLOTS * 8;
                    const uint32_t t4 = rs4_tag(k);
                    const uint4 g = make_uint4(rs4_pack(m[0], t4), rs4_pack(m[1], t4), rs4_pack(m[2], t4), rs4_pack(m[3], t4));
                    if (LOCAL) lv_xcd_store_q4(d, g); else lv_agent_store_q4(d, g);
                } else {
                gran_t* d = dst4 + (long)(2 * n4) * PMEMBERS * SLOTS;
                put(d, rs_pack(m[0], m[1], tag));
                put(d + 4 * RP, rs_pack(m[2], m[3], tag));
                }
            } else if constexpr (QB) {
                if ((l & 15) < RP) {
                    const uint32_t t4 = rs4_tag(k);
                    char* d0 = tx_q + (long)(k & 1) * px_par * 8;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int nb = 4 * n4 + j;
                        char* d = d0 + (long)(nb >> 1) * PMEMBERS * SLOTS * 8 + (nb & 1) * RP * 4 * 16;
                        const uint4 g = make_uint4(rs4_pack(acc[j][0], t4), rs4_pack(acc[j][1], t4), rs4_pack(acc[j][2], t4), rs4_pack(acc[j][3], t4));
                        if (LOCAL) lv_xcd_store_q4(d, g); else lv_agent_store_q4(d, g);
                    }
                }
            } else if ((l & 15) < RP) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int nb = 4 * n4 + j;
                    gran_t* d = dst + (long)(nb >> 1) * PMEMBERS * SLOTS + (nb & 1) * 8 * RP;
                    put(d, rs_pack(acc[j][0], acc[j][1], tag));
                    put(d + 4 * RP, rs_pack(acc[j][2], acc[j][3], tag));
                }
            }
        };
        auto results = [&](f32x4 (&acc)[4]) {
#pragma unroll
            for (int j = 0; j < 4; ++j) LV_MFMA_RESULT(acc[j]);
        };
        f32x4 acc_a[4], acc_b[4], acc_c[4], acc_d[4];
        chunk_mfma(0, acc_a);
        chunk_mfma(1, acc_b);
        results(acc_a);                 // 16 MFMAs behind its last write
        chunk_send(0, acc_a);
        chunk_mfma(2, acc_c);
        results(acc_b);
        chunk_send(1, acc_b);
        chunk_mfma(3, acc_d);
        results(acc_c);
        chunk_send(2, acc_c);
        LV_MFMA_DRAIN();
        results(acc_d);
        chunk_send(3, acc_d);
    };

    bool aborted = false;
    for (int t_hi = T - 1; t_hi >= 0; t_hi -= SBK) {
#pragma unroll
        for (int s2 = 0; s2 < SBK; ++s2) {
            const int t = t_hi - s2;
            if (t < 0 || aborted) continue;       // (no early exit from the unrolled block: its register arrays must stay registers)
            float dh_rec[NP];
#pragma unroll
            for (int q = 0; q < NP; ++q) dh_rec[q] = 0.f;
            LV_TRACE_MARK(t, 0);
            if (t < T - 1) receive(T - 1 - t, dh_rec);        // on a timeout s_abort is set: everybody leaves after the barrier below
            LV_TRACE_MARK(t, 1);
            LV_TRACE_ONLY(LV_TRACE_VAL(t, 6, tr_spins[0]); LV_TRACE_VAL(t, 7, tr_spins[1]);)
            const int par = (T - t) & 1;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                float da[4] = {0.f, 0.f, 0.f, 0.f};
                if (own[q]) {
                    float dh = (has_ext ? dhb[q][s2] : 0.f) + dh_rec[q];
                    if (t == T - 1 && p.dh_last) dh += p.dh_last[pidx[q]];
                    const float ig = recb[q][s2].x, fg = recb[q][s2].y, gg = recb[q][s2].z, og_ = recb[q][s2].w;
                    const float tc = (LV_P16_ABL & 4) ? 0.9f * ctb[q][s2] : lv_tanh_fast(ctb[q][s2]);
                    const float dc = dh * og_ * (1.f - tc * tc) + dc_rec[q];
                    const float d_o = dh * tc;
                    const float d_i = dc * gg, d_g = dc * ig, d_f = dc * ctb[q][s2 + 1];
                    da[0] = d_i * ig * (1.f - ig);
                    da[1] = d_f * fg * (1.f - fg);
                    da[2] = d_g * (1.f - gg * gg);
                    da[3] = d_o * og_ * (1.f - og_);
                    dc_rec[q] = dc * fg;
#pragma unroll
                    for (int g = 0; g < 4; ++g) gsum[q][g] += da[g];
                }
                const uint32_t lo = lv_pack_bf16x2(da[0], da[1]), hi = lv_pack_bf16x2(da[2], da[3]);
                if (own[q]) {
                    sm.dgl[par][prow[q] * DP16 + 2 * uw] = lo; sm.dgl[par][prow[q] * DP16 + 2 * uw + 1] = hi;
                    sm.og[par][prow[q]][0][uw] = (uint16_t)(lo & 0xFFFFu); sm.og[par][prow[q]][1][uw] = (uint16_t)(lo >> 16);
                    sm.og[par][prow[q]][2][uw] = (uint16_t)(hi & 0xFFFFu); sm.og[par][prow[q]][3][uw] = (uint16_t)(hi >> 16);
                }
            }
            // Bulk input of the NEXT block right behind a completed receive (and the last use of this block's registers): a wave's
            // loads retire in order, so anything issued just before a poll is waited for by that poll with its full HBM latency
            // (the first build loaded at the block boundary, i.e. in front of the next receive: 3.8 us per boundary at 16 rows).
            if (s2 == SBK - 1) load_block(t_hi - SBK);
            LV_TRACE_MARK(t, 2);
            if (!(LV_P16_ABL & 8))
            __syncthreads();                    // the workgroup's dG image of this step (double-buffered by step parity); compiles to
                                                // lgkmcnt(0) + s_barrier: the next block's loads stay in flight across it
            LV_TRACE_MARK(t, 3);
            if (s_abort) { aborted = true; continue; }
            if (t > 0 || closing) send(par, T - t);
            store_step(par, t);
            LV_TRACE_MARK(t, 4);
        }
        if (aborted) { if (tid == 0) atomicExch(p.status, 200 + (t_hi < 0 ? 0 : t_hi)); return; }
    }

#pragma unroll
    for (int q = 0; q < NP; ++q)
        if (own[q]) {
#pragma unroll
            for (int g = 0; g < 4; ++g) p.dGsum[(long)(b0 + prow[q]) * 4 * PH + (long)g * PH + punit] = gsum[q][g];
        }
    if (closing) {
        float s0[NP];
        const bool fine = receive(T, s0);
        __syncthreads();
        if (!fine || s_abort) { if (tid == 0) atomicExch(p.status, 300); return; }
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (own[q]) {
                if (p.dh0) p.dh0[pidx[q]] = s0[q];
                float dc = dc_rec[q];
                if (p.tanh_init) { const float h0 = p.hs[pidx[q]]; dc += s0[q] * (1.f - h0 * h0); }
                if (p.dc0) p.dc0[pidx[q]] = dc;
            }
    } else {
#pragma unroll
        for (int q = 0; q < NP; ++q)
            if (own[q] && p.dc0) p.dc0[pidx[q]] = dc_rec[q];
    }
}

constexpr long XCH_FWD16_BYTES = 2L * PGROUPS * 16 * (PH / 2) * 8;
constexpr long XCH_RS16_BYTES = 2L * PGROUPS * PMEMBERS * PMEMBERS * RS16_SLOTS_MAX * 8;

int check_R(int B, int R) { return R >= 1 && R <= 16 && (long)R * PGROUPS >= B; }

}  // namespace

// exchange buffer: [forward half 0 | forward half 1 | BPTT half 0 | BPTT half 1] (+ slack).  flags bit 1 of the two launches: the
// caller alternates the halves -- bit 2 says which one this launch uses -- and every launch zeroes the other half of its own kind
// in its prologue, so no memset launch precedes it (4 x (4.8 us + a kernel boundary) per training step, profiles/r04z_lstm_fixed_cost.txt).
// The buffer must be zero when first used, and the half named by bit 2 must have been cleared by the previous launch of that kind
// (alternate strictly; lv_lstm_persist16_xch_clear zeroes everything, e.g. after the bit has been used inconsistently).
// Without bit 1 a launch uses half 0 behind a hipMemsetAsync, as before (hipGraph capture: a replay cannot alternate).
constexpr long XCH_FWD_OFF[2] = {0, XCH_FWD16_BYTES};
constexpr long XCH_BWD_OFF[2] = {2 * XCH_FWD16_BYTES, 2 * XCH_FWD16_BYTES + XCH_RS16_BYTES};
extern "C" long lv_lstm_persist16_xch_floats(void) { return (2 * XCH_FWD16_BYTES + 2 * XCH_RS16_BYTES) / 4 + 64; }
extern "C" int lv_lstm_persist16_xch_clear(float* xch, void* stream) {
    if (!xch) return LV_ERR_ARG;
    return (int)hipMemsetAsync(xch, 0, (size_t)(2 * XCH_FWD16_BYTES + 2 * XCH_RS16_BYTES), (hipStream_t)stream);
}
// floats of ONE packed bf16 image of W_hh (forward or BPTT form: 128 waves x 64 fragments x 64 lanes x 16 bytes = 8 MB)
extern "C" long lv_lstm_persist16_wpk_floats(void) { return 128L * 64 * 64 * 16 / 4; }

// floats of the saved-activation buffer the 16-row forward writes and the 16-row BPTT reads (workgroup-major, see the top of the
// file): T timesteps at R rows per XCD group.  Both calls must be given the same T, B and R.
extern "C" long lv_lstm_persist16_saved_floats(int T, int R) {
    return T < 0 || R < 1 || R > 16 ? 0 : (long)PGROUPS * PMEMBERS * T * R * SAVED_PER_ROW;
}

namespace {
// canonical saved activations (gate records [T][B][H][4], cell states cs [T+1][B][H]: what the launch-per-timestep forward
// kernels write) -> the workgroup-major record buffer the persistent BPTT reads.  One thread per (t, b, unit).
__global__ __launch_bounds__(256) void import_saved16_kernel(const float* __restrict__ gates, const float* __restrict__ cs,
                                                             float* __restrict__ saved, int T, int B, int R) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long BH = (long)B * PH;
    if (idx >= (long)T * BH) return;
    const int u = (int)(idx % PH), b = (int)((idx / PH) % B), t = (int)(idx / BH);
    const int group = b / R, row = b % R, member = u >> 5, uw = u & 31;
    const long rec = (long)R * SAVED_PER_ROW;
    float* dst = saved + ((long)(group * PMEMBERS + member) * T + t) * rec;
    const f32x4 g = *reinterpret_cast<const f32x4*>(gates + idx * 4);
    *reinterpret_cast<f32x4*>(dst + (row * 32 + uw) * 4) = g;
    dst[R * 128 + row * 32 + uw] = cs[idx + BH];          // record t carries c_t = cs[t + 1]
}
}  // namespace

// Saved activations of a forward that ran on the launch-per-timestep kernels (lv_lstm_fwd_f32 / _bf16: gates [T][B][H][4], cs
// [T+1][B][H]) -> the record buffer of lv_lstm_bwd_bf16_persist16 (lv_lstm_persist16_saved_floats(T, R) floats) for the same
// T, B, R.  Lets an exact-f32 forward recurrence (the KL of encoder.py:55 depends on its last state alone) be followed by the
// persistent BPTT.
extern "C" int lv_lstm_persist16_import_saved(const float* gates, const float* cs, float* saved, int T, int B, int R, int H,
                                              void* stream) {
    if (!gates || !cs || !saved) return LV_ERR_ARG;
    if (T < 0 || B <= 0) return LV_ERR_SHAPE;
    if (H != PH || !check_R(B, R)) return LV_ERR_UNSUPPORTED;
    if (((((uintptr_t)gates) | ((uintptr_t)saved)) & 15) != 0) return LV_ERR_ALIGN;
    if (T == 0) return LV_OK;
    LV_LAUNCH(import_saved16_kernel, dim3((unsigned)lv_cdiv((long)T * B * PH, 256)), dim3(256), 0, stream, gates, cs, saved, T, B, R);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// W_hh [4H][H] f32 -> the register image of lv_lstm_fwd_bf16_persist16 (backward = 0) / lv_lstm_bwd_bf16_persist16 (backward = 1):
// lv_lstm_persist16_wpk_floats() floats, 16-byte aligned.  Re-run only when the weights change.
extern "C" int lv_lstm_persist16_pack(const float* whh, float* wpk, int backward, int H, void* stream) {
    if (!whh || !wpk) return LV_ERR_ARG;
    if (H != PH) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0) return LV_ERR_ALIGN;
    const dim3 grid((unsigned)lv_cdiv(128L * 64 * 64, 256)), block(256);
    if (backward == 1) LV_LAUNCH(pack_w_rs16_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk));
    else LV_LAUNCH(pack_w_k16_kernel, grid, block, 0, stream, whh, reinterpret_cast<uint4*>(wpk), backward == 2 ? 1 : 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Both register images in one launch (forward -> wpk_fwd, BPTT -> wpk_bwd; each lv_lstm_persist16_wpk_floats() floats).
extern "C" int lv_lstm_persist16_pack2(const float* whh, float* wpk_fwd, float* wpk_bwd, int H, void* stream) {
    if (!whh || !wpk_fwd || !wpk_bwd) return LV_ERR_ARG;
    if (H != PH) return LV_ERR_UNSUPPORTED;
    if (((((uintptr_t)wpk_fwd) | ((uintptr_t)wpk_bwd)) & 15) != 0) return LV_ERR_ALIGN;
    LV_LAUNCH(pack_w_both16_kernel, dim3((unsigned)lv_cdiv(2 * 128L * 64 * 64, 256)), dim3(256), 0, stream, whh,
              reinterpret_cast<uint4*>(wpk_fwd), reinterpret_cast<uint4*>(wpk_bwd), 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
// ... with the FORWARD image in IEEE binary16 (for lv_lstm_fwd_bf16_persist16 with flags bit 5) and the BPTT image in bf16
extern "C" int lv_lstm_persist16_pack2_h16(const float* whh, float* wpk_fwd, float* wpk_bwd, int H, void* stream) {
    if (!whh || !wpk_fwd || !wpk_bwd) return LV_ERR_ARG;
    if (H != PH) return LV_ERR_UNSUPPORTED;
    if (((((uintptr_t)wpk_fwd) | ((uintptr_t)wpk_bwd)) & 15) != 0) return LV_ERR_ALIGN;
    LV_LAUNCH(pack_w_both16_kernel, dim3((unsigned)lv_cdiv(2 * 128L * 64 * 64, 256)), dim3(256), 0, stream, whh,
              reinterpret_cast<uint4*>(wpk_fwd), reinterpret_cast<uint4*>(wpk_bwd), 1);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Forward recurrence in one persistent launch with R batch rows per XCD group (1 <= R <= 16, 8 R >= B): groups
// [0, ceil(B / R)) carry the batch, the workgroups of the other groups return at once -- R = 8 at B = 32 runs the recurrence on
// four XCDs and leaves the other four to concurrent kernels.  gx: unit-major input projections [T][B][4H]; wpk: forward image of
// lv_lstm_persist16_pack; hs [T+1][B][H] (slot 0 = initial state, read); no in-kernel dropout (the engine applies dropout_out while h
// is converted to its bf16 images); the saved activations go to the
// workgroup-major buffer of lv_lstm_persist16_saved_floats(T, R) floats instead of gates / cs[1 .. T - 1] (cs: slot 0 is read,
// slot T written); exchange buffer of lv_lstm_persist16_xch_floats() floats.  flags bit 0: hand-off stores without the agent-scope write-through (they stay in the
// XCD's L2; correct while every group is XCD-local -- the round-robin placement of a 256-CU device -- and reported through
// *status as a hand-off timeout otherwise).
extern "C" int lv_lstm_fwd_bf16_persist16(const float* gx, const float* wpk, float* hs, float* cs, float* saved, float* xch,
                                          int* status, int T, int B, int R, int flags, int H, void* stream) {
    if (!gx || !wpk || !hs || !cs || !saved || !xch || !status) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (H != PH || !check_R(B, R)) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)saved) & 15) != 0 || (((uintptr_t)gx) & 15) != 0 || (((uintptr_t)xch) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;      // the groups in use must be resident at once
    if (T == 0) return LV_OK;
    char* const xb = reinterpret_cast<char*>(xch);
    const int dbl = (flags >> 1) & 1, half = dbl ? (flags >> 2) & 1 : 0;
    gran_t* hx = reinterpret_cast<gran_t*>(xb + XCH_FWD_OFF[half]);
    if (!dbl) (void)hipMemsetAsync(hx, 0, (size_t)XCH_FWD16_BYTES, (hipStream_t)stream);
    Fwd16P p{gx, reinterpret_cast<const uint4*>(wpk), hs, cs, saved, hx, status, T, B, R,
             dbl ? reinterpret_cast<uint4*>(xb + XCH_FWD_OFF[1 - half]) : nullptr, XCH_FWD16_BYTES / 16};
    const dim3 grid(PGROUPS * PMEMBERS), block(256);
    if (flags & 32) {                // binary16 recurrent operands (wpk packed by lv_lstm_persist16_pack(.., 2, ..) / _pack2_h16)
        if (flags & 1) {
            if (R <= 4) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<4, true, true>), grid, block, 0, stream, p);
            else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<8, true, true>), grid, block, 0, stream, p);
            else LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<16, true, true>), grid, block, 0, stream, p);
        } else {
            if (R <= 4) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<4, false, true>), grid, block, 0, stream, p);
            else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<8, false, true>), grid, block, 0, stream, p);
            else LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<16, false, true>), grid, block, 0, stream, p);
        }
        LV_CHECK_LAUNCH();
        return LV_OK;
    }
    if (flags & 1) {
        if (R <= 4) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<4, true>), grid, block, 0, stream, p);
        else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<8, true>), grid, block, 0, stream, p);
        else LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<16, true>), grid, block, 0, stream, p);
    } else {
        if (R <= 4) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<4, false>), grid, block, 0, stream, p);
        else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<8, false>), grid, block, 0, stream, p);
        else LV_LAUNCH_RESIDENT((lstm_fwd_persist_k16_kernel<16, false>), grid, block, 0, stream, p);
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// BPTT in one persistent launch, R batch rows per XCD group (as above).  dh_ext [T][B][H] (external gradient of every h_t, or
// NULL) / dh_last [B][H] (of h_T, or NULL); no in-kernel dropout mask; image-only (dG16 = bf16 gate gradients [T][B][4H], gate-major).  saved: as the forward with the same T, B, R wrote it; cs: slot 0 only.
extern "C" int lv_lstm_bwd_bf16_persist16(const float* dh_ext, const float* dh_last, const float* wpk, const float* saved,
                                          const float* hs, const float* cs, uint16_t* dG16, float* dGsum, float* xch, int* status,
                                          float* dh0, float* dc0, int tanh_init, int T, int B, int R, int flags, int H, void* stream) {
    if (!wpk || !saved || !cs || !dG16 || !dGsum || !xch || !status) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || H <= 0) return LV_ERR_SHAPE;
    if (tanh_init && !hs) return LV_ERR_ARG;
    if (H != PH || !check_R(B, R)) return LV_ERR_UNSUPPORTED;
    if ((((uintptr_t)wpk) & 15) != 0 || (((uintptr_t)saved) & 15) != 0 || (((uintptr_t)xch) & 15) != 0 || (((uintptr_t)dG16) & 15) != 0)
        return LV_ERR_ALIGN;
    if (lv_device_cus() < PGROUPS * PMEMBERS) return LV_ERR_UNSUPPORTED;
    char* const xb = reinterpret_cast<char*>(xch);
    const int dbl = (flags >> 1) & 1, half = dbl ? (flags >> 2) & 1 : 0;
    gran_t* gxch = reinterpret_cast<gran_t*>(xb + XCH_BWD_OFF[half]);
    const int RPi = R <= 4 ? 4 : (R <= 8 ? 8 : 16);
    const long extent = XCH_RS16_BYTES / RS16_SLOTS_MAX * 16 * RPi;                  // the instantiation's dense extent
    // double-buffered: the other half is cleared over the extent of the launch that last USED it -- flags bits 3..4 name that
    // launch's instantiation (1 / 2 / 3 = 4 / 8 / 16 rows; 0 = the same as this one): a batch size that changes between launches
    // changes the extent, and stale tags beyond a smaller clear would be read as data
    const int ccls = (flags >> 3) & 3;
    const long cextent = XCH_RS16_BYTES / RS16_SLOTS_MAX * 16 * (ccls == 0 ? RPi : (4 << (ccls - 1)));
    if (!dbl) (void)hipMemsetAsync(gxch, 0, (size_t)extent, (hipStream_t)stream);
    Bwd16P p{dh_ext, dh_last, reinterpret_cast<const uint4*>(wpk), saved, cs, hs, dG16, dGsum, dh0, dc0, tanh_init, gxch, status, T, B, R,
             dbl ? reinterpret_cast<uint4*>(xb + XCH_BWD_OFF[1 - half]) : nullptr, cextent / 16};
    const dim3 grid(PGROUPS * PMEMBERS), block(256);
    if (flags & 1) {
        if (R <= 4) LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<4, true>), grid, block, 0, stream, p);
        else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<8, true>), grid, block, 0, stream, p);
        else LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<16, true>), grid, block, 0, stream, p);
    } else {
        if (R <= 4) LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<4, false>), grid, block, 0, stream, p);
        else if (R <= 8) LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<8, false>), grid, block, 0, stream, p);
        else LV_LAUNCH_RESIDENT((lstm_bwd_persist_rs16_kernel<16, false>), grid, block, 0, stream, p);
    }
    LV_CHECK_LAUNCH();
    return LV_OK;
}
