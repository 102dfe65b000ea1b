// lv_head.hip -- the batch-sized ends of the two LSTM networks, each as ONE launch instead of a chain of tiny GEMMs.
//
// Around the two recurrences the reference runs a handful of contractions whose "long" dimension is the batch or the
// latent size: the encoder head h_T W_lin^T + reparameterise + KL (modules/encoders/enc_lstm.py:62, encoder.py:53-79),
// the decoder's initial state c0 = trans_linear(z), h0 = tanh(c0) and the z-part of its input projection
// (modules/decoders/dec_lstm.py:95-101), and their backward (text.py:384).  At B = 32, nz = 32 each of them is a few
// MFLOP: as separate tile GEMMs they cost 6-36 us apiece (a 128x128 tile kernel on a 32 x 32 x 4096 product) plus a
// launch boundary each.  Here every output element is one short dot product computed by one thread (or one wave), with
// the small operand staged in LDS.  Arithmetic is f32 FMA chains in a fixed order: deterministic, within 1e-6 of the
// MFMA GEMM route.
#include "lv_device.h"

namespace {

// Staging loop with its loads IN FLIGHT TOGETHER: element i of n comes from src(i) -- a valid address for every i < n -- and goes to
// put(i, value).  Eight elements per thread and round trip, the loads from clamped indices and unconditional; written as a plain
// `for (i = tid; i < n; i += 256) put(i, *src(i))` hipcc keeps one load -> wait -> LDS store per iteration whenever the trip count is
// a runtime value (dec_tail_bwd_kernel: 8 + 2 + 1 dependent memory round trips in its three staging loops, half of its 21 us).
// sum_i a(i) b(i) as TWO fma chains (even i, odd i) added at the end -- the order every dot product of this file has used -- with the
// factors of eight terms read (from LDS) before their fmas: a runtime-length loop of `read, read, fma` waits for the LDS after every
// pair (dec_tail_bwd_kernel: 2 x 6 us of LDS latency in its two contraction loops).
template <class FA, class FB> __device__ __forceinline__ float dot2_batched(int n, FA a, FB b) {
    float s0 = 0.f, s1 = 0.f;
    int i = 0;
    for (; i + 7 < n; i += 8) {
        float x[8], y[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { x[u] = a(i + u); y[u] = b(i + u); }
#pragma unroll
        for (int u = 0; u < 8; u += 2) { s0 = fmaf(x[u], y[u], s0); s1 = fmaf(x[u + 1], y[u + 1], s1); }
    }
    for (; i + 1 < n; i += 2) { s0 = fmaf(a(i), b(i), s0); s1 = fmaf(a(i + 1), b(i + 1), s1); }
    if (i < n) s0 = fmaf(a(i), b(i), s0);
    return s0 + s1;
}

template <class Src, class Put> __device__ __forceinline__ void stage_batched(int n, int tid, Src src, Put put) {
    for (int i0 = tid; i0 < n; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; v[u] = *src(i < n ? i : n - 1); }
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + 256 * u; if (i < n) put(i, v[u]); }
    }
}

// ---- encoder head forward: mulv = h_T . W_lin^T ; z = mu + eps*exp(lv/2) ; KL = 0.5*sum(mu^2 + exp(lv) - lv - 1) -------
// one workgroup per batch row; a wave owns outputs o = w, w+4, ... and keeps 16 of them in flight (independent loads)
__global__ __launch_bounds__(256) void enc_head_fwd_kernel(const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           const float* __restrict__ eps, float* __restrict__ mulv,
                                                           float* __restrict__ z, float* __restrict__ kl,
                                                           int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    float* sh = reinterpret_cast<float*>(smem);          // [H] the row of h_T
    float* sm = sh + H;                                   // [2nz] mu | logvar
    float* sk = sm + 2 * nz;                              // [nz] KL terms
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    const int nz2 = 2 * nz;
    for (int h = tid; h < H; h += 256) sh[h] = hT[(long)b * H + h];
    __syncthreads();
    for (int o0 = 0; o0 < nz2; o0 += 64) {
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll 4
        for (int h = l; h < H; h += 64) {                 // (4 x 16 independent loads in flight: the loop is a chain of L2 round trips)
            const float x = sh[h];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int o = o0 + w + 4 * i;
                acc[i] = fmaf(x, wlin[(long)(o < nz2 ? o : 0) * H + h], acc[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int o = o0 + w + 4 * i;
            const float s = lv_wave_sum(acc[i]);
            if (l == 0 && o < nz2) { sm[o] = s; mulv[(long)b * nz2 + o] = s; }
        }
    }
    __syncthreads();
    for (int j = tid; j < nz; j += 256) {
        const float m = sm[j], lv = sm[nz + j];
        const float sd = expf(0.5f * lv);
        sk[j] = (m * m + expf(lv)) - lv - 1.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)b * ns + s) * nz + j;
            z[zi] = m + eps[zi] * sd;
        }
    }
    __syncthreads();
    if (tid == 0) {
        float s = 0.f;
        for (int j = 0; j < nz; ++j) s += sk[j];
        kl[b] = 0.5f * s;
    }
}

// ---- encoder head backward: dmulv from (dz, dKL); dh_T = dmulv . W_lin ; dW_lin = dmulv^T . h_T --------------------------
// grid ceil(H/64): a workgroup covers 64 columns h.  Everything it needs -- dmulv [B][2nz], the W_lin columns [2nz][64]
// and the h_T columns [B][64] -- is staged in LDS with fully parallel loads first; the dot products then run out of LDS
// (a thread that walks a runtime-length loop of global load -> fma pairs pays one memory round trip per iteration: the
// first versions of these kernels took 60-250 us that way).  dz may arrive as `parts` partial sums [parts][B][ns][nz].
template <int HB_COLS>
__global__ __launch_bounds__(256) void enc_head_bwd_kernel(const float* __restrict__ mulv, const float* __restrict__ eps,
                                                           const float* __restrict__ dz, int parts, const float* __restrict__ dkl,
                                                           const float* __restrict__ hT, const float* __restrict__ wlin,
                                                           float* __restrict__ dmulv, float* __restrict__ dhT,
                                                           float* __restrict__ gwlin, int B, int H, int ns, int nz) {
    LV_DYN_SHARED(smem);
    const int nz2 = 2 * nz;
    float* sd = reinterpret_cast<float*>(smem);           // [B][2nz]   dmulv
    float* sw = sd + B * nz2;                             // [2nz][64]  W_lin columns of this block
    float* shh = sw + nz2 * HB_COLS;                      // [B][64]    h_T columns of this block
    const int tid = (int)threadIdx.x;
    const int h0 = (int)blockIdx.x * HB_COLS;
    const long pstride = (long)B * ns * nz;
    // staging loads are UNCONDITIONAL (clamped column, the value dropped by a select afterwards): a load behind a condition is
    // waited for right behind its issue, which turns a staging loop into a chain of memory round trips
    stage_batched(nz2 * HB_COLS, tid,
                  [&](int i) { const int j = i / HB_COLS, c = i % HB_COLS; return wlin + (long)j * H + (h0 + c < H ? h0 + c : H - 1); },
                  [&](int i, float v) { sw[i] = h0 + i % HB_COLS < H ? v : 0.f; });
    stage_batched(B * HB_COLS, tid,
                  [&](int i) { const int bb = i / HB_COLS, c = i % HB_COLS; return hT + (long)bb * H + (h0 + c < H ? h0 + c : H - 1); },
                  [&](int i, float v) { shh[i] = h0 + i % HB_COLS < H ? v : 0.f; });
    // dz = sum of the partial sums (in order), staged in LDS first
    float* sg = shh + B * HB_COLS;                         // [B*ns*nz] summed dz
    const int ne = B * ns * nz;
    // (two elements x 32 parts = 64 independent loads per thread and round trip: with 8 in flight the 80 parts of the decoder tail
    //  were 40 dependent round trips per thread, 20 of this kernel's 30 us)
    // 16-byte form (four consecutive elements per thread and part, 16 parts in flight): a quarter of the load instructions -- the
    // address unit takes a quad of lanes per cycle whatever the width, and the compiler kept only 16-18 of the 64 scalar loads of a
    // batch in flight.  Same order of addition per element.
    const bool v4 = (ne & 3) == 0 && (pstride & 3) == 0 && (((uintptr_t)dz) & 15) == 0;
    if (v4) {
        for (int e = 4 * tid; e < ne; e += 1024) {
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q0 = 0; q0 < parts; q0 += 16) {
                float4 pv[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) pv[u] = *reinterpret_cast<const float4*>(dz + (long)(q0 + u < parts ? q0 + u : 0) * pstride + e);
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (q0 + u < parts) { g.x += pv[u].x; g.y += pv[u].y; g.z += pv[u].z; g.w += pv[u].w; }
            }
            sg[e] = g.x; sg[e + 1] = g.y; sg[e + 2] = g.z; sg[e + 3] = g.w;
        }
    } else
    for (int e = tid; e < ne; e += 512) {
        const int e1 = e + 256 < ne ? e + 256 : e;
        float g0 = 0.f, g1 = 0.f;
        for (int q0 = 0; q0 < parts; q0 += 32) {
            float pa[32], pb[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const long off = (long)(q0 + u < parts ? q0 + u : 0) * pstride;
                pa[u] = dz[off + e];
                pb[u] = dz[off + e1];
            }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                g0 += q0 + u < parts ? pa[u] : 0.f;
                g1 += q0 + u < parts ? pb[u] : 0.f;
            }
        }
        sg[e] = g0;
        if (e + 256 < ne) sg[e + 256] = g1;
    }
    __syncthreads();
    for (int i = tid; i < B * nz; i += 256) {
        const int bb = i / nz, j = i % nz;
        const float m = mulv[(long)bb * nz2 + j], lv = mulv[(long)bb * nz2 + nz + j];
        const float sdv = expf(0.5f * lv);
        float gz = 0.f, gze = 0.f;
        for (int s = 0; s < ns; ++s) {
            const long zi = ((long)bb * ns + s) * nz + j;
            const float g = sg[zi];
            gz += g;
            gze += g * eps[zi];
        }
        const float gk = dkl[bb];
        const float dm = gz + gk * m, dl = gze * (0.5f * sdv) + gk * (0.5f * (expf(lv) - 1.f));
        sd[bb * nz2 + j] = dm;
        sd[bb * nz2 + nz + j] = dl;
        if (blockIdx.x == 0) { dmulv[(long)bb * nz2 + j] = dm; dmulv[(long)bb * nz2 + nz + j] = dl; }
    }
    __syncthreads();
    const int c = tid % HB_COLS, g4 = tid / HB_COLS;
    if (h0 + c >= H) return;
    for (int r = g4; r < B + nz2; r += 256 / HB_COLS) {
        if (r < B) {                                      // dh_T[r][h] = sum_j dmulv[r][j] W_lin[j][h]
            dhT[(long)r * H + h0 + c] = dot2_batched(nz2, [&](int j) { return sd[r * nz2 + j]; }, [&](int j) { return sw[j * HB_COLS + c]; });
        } else {                                          // dW_lin[j][h] = sum_b dmulv[b][j] h_T[b][h]
            const int j = r - B;
            gwlin[(long)j * H + h0 + c] = dot2_batched(B, [&](int bb) { return sd[bb * nz2 + j]; }, [&](int bb) { return shh[bb * HB_COLS + c]; });
        }
    }
}

// ---- decoder initial state and the z-part of its input projection ------------------------------------------------------
// rows n < H of the stacked weight [W_trans ; W_ih[:, col0:]]: c0[b][n] = z[b] . W_trans[n], h0 = tanh(c0); rows H + n', n' < 4H:
// Zp[b][n'] = z[b] . W_ih[n'][col0:] + b_ih + b_hh (written gate-major, or unit-major 4u+g when unit_major != 0).
// A workgroup takes DI_ROWS stacked rows: their weights ([rows][nz]) and z ([B][nz]) are staged in LDS with parallel loads;
// thread = (row, batch quarter) then runs nz-long dots out of LDS (row pitch nz + 1: conflict-free across rows).
constexpr int DI_ROWS = 64;
__global__ __launch_bounds__(256) void dec_init_kernel(const float* __restrict__ z, const float* __restrict__ wtr,
                                                       const float* __restrict__ wih, long ld_wih, int col0,
                                                       const float* __restrict__ bih, const float* __restrict__ bhh,
                                                       float* __restrict__ c0, float* __restrict__ h0, float* __restrict__ zp,
                                                       int unit_major, int B, int H, int nz) {
    LV_DYN_SHARED(smem);
    const int pw = nz + 1;
    float* sz = reinterpret_cast<float*>(smem);           // [B][nz]
    float* swt = sz + B * nz;                             // [DI_ROWS][nz + 1]
    const int tid = (int)threadIdx.x;
    const int n0 = (int)blockIdx.x * DI_ROWS;
    stage_batched(B * nz, tid, [&](int i) { return z + i; }, [&](int i, float v) { sz[i] = v; });
    stage_batched(DI_ROWS * nz, tid,
                  [&](int i) {
                      const int rr = i / nz, k = i % nz, n = n0 + rr;
                      const int nc = n < 5 * H ? n : 5 * H - 1;          // rows beyond the stack read its last row (dropped below)
                      return nc < H ? wtr + (long)nc * nz + k : wih + (long)(nc - H) * ld_wih + col0 + k;
                  },
                  [&](int i, float v) { const int rr = i / nz, k = i % nz; swt[rr * pw + k] = n0 + rr < 5 * H ? v : 0.f; });
    __syncthreads();
    const int rr = tid & 63, bq = tid >> 6;
    const int n = n0 + rr;
    if (n >= 5 * H) return;
    const bool init = n < H;
    const int r = init ? n : n - H;
    const float bias = init ? 0.f : bih[r] + bhh[r];
    long ocol = r;
    if (!init && unit_major) ocol = 4L * (r % H) + r / H;
    const float* wr = swt + rr * pw;
    for (int b = bq; b < B; b += 4) {
        const float* zb = sz + b * nz;
        const float sv = dot2_batched(nz, [&](int k) { return zb[k]; }, [&](int k) { return wr[k]; });
        if (init) { c0[(long)b * H + r] = sv; h0[(long)b * H + r] = tanhf(sv); }
        else zp[(long)b * 4 * H + ocol] = sv + bias;
    }
}

// ---- decoder tail of the backward ---------------------------------------------------------------------------------------
// Stacked contraction rows: row < 4H is gate row n (d = dGsum[:, n], weights W_ih[n][col0:]), row 4H + j is unit j
// (d = dc0[:, j], weights W_trans[j]).  Workgroup `part` takes DZ_ROWS of them, stages d [B][rows], the weights [rows][nz]
// and z [B][nz] in LDS with parallel loads, then
//   * weight gradients: out[row][k] = sum_b d[b][row] z[b][k] (gW_ih[:, col0:] / gW_trans) -- lanes run over k, so a row's nz
//     outputs are one contiguous store -- and the bias gradients g_bih[n] = g_bhh[n] = sum_b d[b][n];
//   * its partial of dz: dzp[part][b][k] = sum_{rows} d[b][row] W[row][k]   (the consumer sums the parts in order).
constexpr int DZ_ROWS = 64;
__global__ __launch_bounds__(256) void dec_tail_bwd_kernel(const float* __restrict__ dGsum, const float* __restrict__ dc0,
                                                           const float* __restrict__ z, const float* __restrict__ wih,
                                                           long ld_wih, int col0, const float* __restrict__ wtr,
                                                           float* __restrict__ gwih, long ld_gwih, float* __restrict__ gwtr,
                                                           float* __restrict__ gbih, float* __restrict__ gbhh,
                                                           float* __restrict__ dzp, int B, int H, int nz) {
    LV_DYN_SHARED(smem);
    const int pd = DZ_ROWS + 1;
    float* sz = reinterpret_cast<float*>(smem);           // [B][nz]
    float* sdv = sz + B * nz;                             // [B][DZ_ROWS + 1]
    float* swt = sdv + B * pd;                            // [DZ_ROWS][nz]
    const int tid = (int)threadIdx.x;
    const int part = (int)blockIdx.x;
    const int row0 = part * DZ_ROWS;
    const int nrows = 5 * H - row0 < DZ_ROWS ? 5 * H - row0 : DZ_ROWS;
    stage_batched(B * nz, tid, [&](int i) { return z + i; }, [&](int i, float v) { sz[i] = v; });
    stage_batched(B * DZ_ROWS, tid,
                  [&](int i) {
                      const int b = i / DZ_ROWS, rr = i % DZ_ROWS;
                      const int row = rr < nrows ? row0 + rr : row0;     // rows beyond the slice read its first row (dropped below)
                      return row < 4 * H ? dGsum + (long)b * 4 * H + row : dc0 + (long)b * H + (row - 4 * H);
                  },
                  [&](int i, float v) { const int b = i / DZ_ROWS, rr = i % DZ_ROWS; sdv[b * pd + rr] = rr < nrows ? v : 0.f; });
    stage_batched(DZ_ROWS * nz, tid,
                  [&](int i) {
                      const int rr = i / nz, k = i % nz;
                      const int row = rr < nrows ? row0 + rr : row0;
                      return row < 4 * H ? wih + (long)row * ld_wih + col0 + k : wtr + (long)(row - 4 * H) * nz + k;
                  },
                  [&](int i, float v) { swt[i] = (i / nz) < nrows ? v : 0.f; });
    __syncthreads();
    // weight gradients: (row, k) pairs dealt to the threads with k fastest
    for (int i = tid; i < nrows * nz; i += 256) {
        const int rr = i / nz, k = i % nz, row = row0 + rr;
        const float sv = dot2_batched(B, [&](int b) { return sdv[b * pd + rr]; }, [&](int b) { return sz[b * nz + k]; });
        if (row < 4 * H) gwih[(long)row * ld_gwih + col0 + k] = sv;
        else gwtr[(long)(row - 4 * H) * nz + k] = sv;
    }
    for (int rr = tid; rr < nrows; rr += 256) {           // bias gradients (gate rows only)
        const int row = row0 + rr;
        if (row < 4 * H) {
            float t = 0.f;
            for (int b = 0; b < B; ++b) t += sdv[b * pd + rr];
            gbih[row] = t; gbhh[row] = t;
        }
    }
    // partial dz: (b, k) pairs dealt to the threads with k fastest
    for (int i = tid; i < B * nz; i += 256) {
        const int b = i / nz, k = i % nz;
        dzp[((long)part * B + b) * nz + k] = dot2_batched(nrows, [&](int rr) { return sdv[b * pd + rr]; }, [&](int rr) { return swt[rr * nz + k]; });
    }
}

}  // namespace

// LSTMEncoder head + GaussianEncoderBase.encode (enc_lstm.py:62-64, encoder.py:40-57) in one launch.
// hT [B][H] (last hidden state), W_lin [2nz][H], eps [B][ns][nz] -> mulv [B][2nz], z [B][ns][nz], kl [B]
extern "C" int lv_enc_head_fwd_f32(const float* hT, const float* wlin, const float* eps, float* mulv, float* z, float* kl,
                                   int B, int H, int ns, int nz, void* stream) {
    if (!hT || !wlin || !eps || !mulv || !z || !kl) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || ns <= 0 || nz <= 0) return LV_ERR_SHAPE;
    const size_t sh = (size_t)(H + 3 * nz) * sizeof(float);
    if (sh > 60000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(enc_head_fwd_kernel, dim3((unsigned)B), dim3(256), sh, stream, hT, wlin, eps, mulv, z, kl, H, ns, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Backward of the same: (dz, dkl [B]) -> dmulv [B][2nz], dhT [B][H] (gradient entering the BPTT at the last step),
// gW_lin [2nz][H] ('=' semantics).  dz: [dz_parts][B][ns][nz], summed over the leading index in order (dz_parts = 1 for a
// plain gradient; lv_dec_tail_bwd_f32 hands over lv_dec_tail_parts(H) partial sums).
extern "C" int lv_enc_head_bwd_f32(const float* mulv, const float* eps, const float* dz, int dz_parts, const float* dkl,
                                   const float* hT, const float* wlin, float* dmulv, float* dhT, float* gwlin, int B, int H,
                                   int ns, int nz, void* stream) {
    if (!mulv || !eps || !dz || !dkl || !hT || !wlin || !dmulv || !dhT || !gwlin) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || ns <= 0 || nz <= 0 || dz_parts <= 0) return LV_ERR_SHAPE;
    const size_t base = (size_t)B * 2 * nz + (size_t)B * ns * nz, per_col = (size_t)2 * nz + B;
    if ((base + per_col * 16) * sizeof(float) <= 64000)
        LV_LAUNCH(enc_head_bwd_kernel<16>, dim3((unsigned)lv_cdiv(H, 16)), dim3(256), (base + per_col * 16) * sizeof(float), stream,
                  mulv, eps, dz, dz_parts, dkl, hT, wlin, dmulv, dhT, gwlin, B, H, ns, nz);
    else
        return LV_ERR_UNSUPPORTED;
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// LSTMDecoder.decode's z-dependent prologue (dec_lstm.py:95-101): c0 = z W_trans^T, h0 = tanh(c0), and
// Zp = z W_ih[:, col0:]^T + b_ih + b_hh (the part of the input projection that cat((word_embed, z_)) contributes),
// gate-major [B][4H] or unit-major (column 4u+g) for the unit-major Gx epilogue.
extern "C" int lv_dec_init_f32(const float* z, const float* wtr, const float* wih, long ld_wih, int col0, const float* bih,
                               const float* bhh, float* c0, float* h0, float* zp, int unit_major, int B, int H, int nz,
                               void* stream) {
    if (!z || !wtr || !wih || !bih || !bhh || !c0 || !h0 || !zp) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || nz <= 0 || ld_wih < col0 + nz) return LV_ERR_SHAPE;
    const size_t sh = ((size_t)B * nz + (size_t)DI_ROWS * (nz + 1)) * sizeof(float);
    if (sh > 64000) return LV_ERR_UNSUPPORTED;
    LV_LAUNCH(dec_init_kernel, dim3((unsigned)lv_cdiv(5L * H, DI_ROWS)), dim3(256), sh, stream, z, wtr, wih, ld_wih, col0, bih, bhh,
              c0, h0, zp, unit_major, B, H, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// number of partial sums lv_dec_tail_bwd_f32 writes for dz
extern "C" int lv_dec_tail_parts(int H) { return lv_cdiv(5L * H, DZ_ROWS); }

// Backward of the same, fed by the BPTT's sums: dGsum [B][4H] (gate-major, sum over time of the gate gradients) and
// dc0 [B][H] -> gW_ih[:, col0:col0+nz], g_b_ih, g_b_hh, gW_trans ('=' semantics) and dz = dGsum . W_ih[:, col0:] +
// dc0 . W_trans as lv_dec_tail_parts(H) partial sums dz_parts [parts][B][nz] over slices of the contraction (summed in
// order by the consumer: lv_enc_head_bwd_f32, or lv_colsum_f32 over the leading index).
extern "C" int lv_dec_tail_bwd_f32(const float* dGsum, const float* dc0, const float* z, const float* wih, long ld_wih, int col0,
                                   const float* wtr, float* gwih, long ld_gwih, float* gwtr, float* gbih, float* gbhh,
                                   float* dz_parts, int B, int H, int nz, void* stream) {
    if (!dGsum || !dc0 || !z || !wih || !wtr || !gwih || !gwtr || !gbih || !gbhh || !dz_parts) return LV_ERR_ARG;
    if (B <= 0 || H <= 0 || nz <= 0 || ld_wih < col0 + nz || ld_gwih < col0 + nz) return LV_ERR_SHAPE;
    const size_t sh = ((size_t)B * nz + (size_t)B * (DZ_ROWS + 1) + (size_t)DZ_ROWS * nz) * sizeof(float);
    if (sh > 64000) return LV_ERR_UNSUPPORTED;
    const int parts = lv_cdiv(5L * H, DZ_ROWS);
    LV_LAUNCH(dec_tail_bwd_kernel, dim3((unsigned)parts), dim3(256), sh, stream, dGsum, dc0, z, wih, ld_wih, col0, wtr, gwih,
              ld_gwih, gwtr, gbih, gbhh, dz_parts, B, H, nz);
    LV_CHECK_LAUNCH();
    return LV_OK;
}
