// lv_embed.hip -- embedding row gather (forward) and deterministic sorted-segment scatter-add (backward).
//
// Replaces nn.Embedding forward (modules/encoders/enc_lstm.py:58, modules/decoders/dec_lstm.py:80) fused with
// the decoder's input dropout (dec_lstm.py:81), and aten::embedding_dense_backward reached from text.py:384
// (decoder embedding has padding_idx = V-1, dec_lstm.py:28, SURVEY.md G3: that row's grad stays zero).
//
// ids stay in the reference's batch-first int64 layout [B][ids_stride]; activations are written time-major
// ([T][B][ni], row r = t*B + b) so each LSTM timestep is one contiguous slab.  Dropout keep-masks are taken
// in the reference's batch-first layout [B][T][ni] (that is what replaying torch's RNG produces).
//
// Backward is sort-based instead of atomic so the result is bit-reproducible: one 512-thread workgroup
// radix-sorts (token, row) pairs (stable, 4 bits per pass, counts in LDS), then one workgroup per segment
// head sums its rows in ascending row order.
#include "lv_device.h"

namespace {

__global__ __launch_bounds__(256) void embed_gather_kernel(const float* __restrict__ emb, const int64_t* __restrict__ ids,
                                                           long ids_stride, const uint8_t* __restrict__ mask, float scale,
                                                           float* __restrict__ X, int T, int B, int ni, int V, int vec) {
    // 4 rows per workgroup, 64 threads (one wave) per row
    const int r = (int)blockIdx.x * 4 + ((int)threadIdx.x >> 6);
    const int l = (int)threadIdx.x & 63;
    if (r >= T * B) return;
    const int t = r / B, b = r % B;
    long tok = ids[(long)b * ids_stride + t];
    if (tok < 0) tok = 0;
    if (tok >= V) tok = V - 1;
    const float* src = emb + tok * (long)ni;
    float* dst = X + (long)r * ni;
    const uint8_t* m = mask ? mask + ((long)b * T + t) * ni : nullptr;
    if (vec) {
        for (int k = l * 4; k < ni; k += 256) {
            float4 v = *reinterpret_cast<const float4*>(src + k);
            if (m) {
                v.x = m[k] ? v.x * scale : 0.f;
                v.y = m[k + 1] ? v.y * scale : 0.f;
                v.z = m[k + 2] ? v.z * scale : 0.f;
                v.w = m[k + 3] ? v.w * scale : 0.f;
            }
            *reinterpret_cast<float4*>(dst + k) = v;
        }
    } else {
        for (int k = l; k < ni; k += 64) {
            float v = src[k];
            if (m) v = m[k] ? v * scale : 0.f;
            dst[k] = v;
        }
    }
}

// ---- stable LSD radix sort of (token, row) pairs by token; single 512-thread workgroup ----
constexpr int SORT_THREADS = 512;
constexpr int SORT_WAVES = SORT_THREADS / 64;

__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_tot /*[SORT_WAVES] LDS*/, int* total) {
    // inclusive scan inside each wave with shuffles
    const int tid = (int)threadIdx.x, l = tid & 63, w = tid >> 6;
    int x = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, (unsigned)d, 64);
        if (l >= d) x += y;
    }
    if (l == 63) wave_tot[w] = x;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < SORT_WAVES; ++i) {
        const int wt = wave_tot[i];
        if (i < w) base += wt;
        tot += wt;
    }
    __syncthreads();
    *total = tot;
    return base + x - v;
}

__global__ __launch_bounds__(SORT_THREADS) void token_sort_kernel(const int64_t* __restrict__ ids, long ids_stride, int T, int B,
                                                                  int V, int* out_rows, int* out_tok, int* tmp) {
    __shared__ int counts[16 * SORT_THREADS];
    __shared__ int wave_tot[SORT_WAVES];
    const int tid = (int)threadIdx.x;
    const int N = T * B;
    const int per = (N + SORT_THREADS - 1) / SORT_THREADS;
    const int e0 = tid * per;
    const int e1 = (e0 + per) < N ? (e0 + per) : N;
    int bits = 1;
    while ((1L << bits) < (long)V) ++bits;
    const int npass = (bits + 3) / 4;
    // ping-pong so that the LAST pass lands in (out_tok, out_rows)
    int* kbuf[2] = {tmp, out_tok};
    int* vbuf[2] = {tmp + N, out_rows};
    int cur = (npass & 1) ? 0 : 1;   // buffer holding the input of pass 0 (only used for pass >= 1)
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 4 * pass;
        const int* kin = kbuf[cur];
        const int* vin = vbuf[cur];
        int* kout = kbuf[cur ^ 1];
        int* vout = vbuf[cur ^ 1];
        int hist[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) hist[d] = 0;
        for (int e = e0; e < e1; ++e) {
            int key;
            if (pass == 0) {
                const int t = e / B, b = e % B;
                long tok = ids[(long)b * ids_stride + t];
                if (tok < 0) tok = 0;
                if (tok >= V) tok = V - 1;
                key = (int)tok;
            } else {
                key = kin[e];
            }
            const int dgt = (key >> shift) & 15;
#pragma unroll
            for (int d = 0; d < 16; ++d) hist[d] += (d == dgt) ? 1 : 0;
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) counts[d * SORT_THREADS + tid] = hist[d];
        __syncthreads();
        // exclusive scan of counts in (digit-major, thread-minor) order: thread j owns [16j, 16j+16)
        int loc[16];
        int s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { loc[i] = counts[tid * 16 + i]; s += loc[i]; }
        int total;
        int base = block_exclusive_scan(s, wave_tot, &total);
#pragma unroll
        for (int i = 0; i < 16; ++i) { counts[tid * 16 + i] = base; base += loc[i]; }
        __syncthreads();
        int off[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) off[d] = counts[d * SORT_THREADS + tid];
        for (int e = e0; e < e1; ++e) {
            int key, val;
            if (pass == 0) {
                const int t = e / B, b = e % B;
                long tok = ids[(long)b * ids_stride + t];
                if (tok < 0) tok = 0;
                if (tok >= V) tok = V - 1;
                key = (int)tok;
                val = e;
            } else {
                key = kin[e];
                val = vin[e];
            }
            const int dgt = (key >> shift) & 15;
            int pos = 0;
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (d == dgt) { pos = off[d]; off[d] = pos + 1; }
            kout[pos] = key;
            vout[pos] = val;
        }
        __syncthreads();   // single workgroup: global writes of this pass visible to the next via the barrier
        cur ^= 1;
    }
}

// The same sort with the intermediate passes in LDS (N <= SORT_LDS_MAX elements: 32-bit keys + 16-bit row ids in two ping-pong
// images).  A radix pass scatters every element to its own address; from one workgroup that is ~13 000 four-byte global stores per
// pass, and those -- not the loads -- were the 55 us of token_sort_kernel at the Yahoo shape.  Here only the last pass writes to
// memory.  Same digit order, same stable placement: the output is identical.
constexpr int SORT_LDS_MAX = 8192;
__global__ __launch_bounds__(SORT_THREADS) void token_sort_lds_kernel(const int64_t* __restrict__ ids, long ids_stride, int T, int B,
                                                                      int V, int* out_rows, int* out_tok) {
    __shared__ int counts[16 * SORT_THREADS];
    __shared__ int wave_tot[SORT_WAVES];
    __shared__ int kimg[2][SORT_LDS_MAX];
    __shared__ uint16_t vimg[2][SORT_LDS_MAX];
    const int tid = (int)threadIdx.x;
    const int N = T * B;
    const int per = (N + SORT_THREADS - 1) / SORT_THREADS;
    const int e0 = tid * per;
    const int e1 = (e0 + per) < N ? (e0 + per) : N;
    int bits = 1;
    while ((1L << bits) < (long)V) ++bits;
    const int npass = (bits + 3) / 4;
    // the tokens of this thread's elements into image 1 (what "pass -1" would have written), 16 loads in flight at a time
    for (int c0 = e0; c0 < e1; c0 += 16) {
        long tk[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = c0 + u < e1 ? c0 + u : e1 - 1;
            const int t = e / B, b = e % B;
            tk[u] = ids[(long)b * ids_stride + t];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (c0 + u < e1) {
                kimg[1][c0 + u] = (int)(tk[u] < 0 ? 0 : (tk[u] >= V ? V - 1 : tk[u]));
                vimg[1][c0 + u] = (uint16_t)(c0 + u);
            }
    }
    __syncthreads();
    for (int pass = 0; pass < npass; ++pass) {
        const int shift = 4 * pass;
        const int* kin = kimg[(pass + 1) & 1];           // pass p reads image (p + 1) & 1 (written by pass p - 1), writes image p & 1
        const uint16_t* vin = vimg[(pass + 1) & 1];
        int* kout = kimg[pass & 1];
        uint16_t* vout = vimg[pass & 1];
        const bool last = pass == npass - 1;
        auto key_of = [&](int e) -> int { return kin[e]; };
        int hist[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) hist[d] = 0;
        for (int e = e0; e < e1; ++e) {
            const int dgt = (key_of(e) >> shift) & 15;
#pragma unroll
            for (int d = 0; d < 16; ++d) hist[d] += (d == dgt) ? 1 : 0;
        }
#pragma unroll
        for (int d = 0; d < 16; ++d) counts[d * SORT_THREADS + tid] = hist[d];
        __syncthreads();
        int loc[16];
        int s = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i) { loc[i] = counts[tid * 16 + i]; s += loc[i]; }
        int total;
        int base = block_exclusive_scan(s, wave_tot, &total);
#pragma unroll
        for (int i = 0; i < 16; ++i) { counts[tid * 16 + i] = base; base += loc[i]; }
        __syncthreads();
        int off[16];
#pragma unroll
        for (int d = 0; d < 16; ++d) off[d] = counts[d * SORT_THREADS + tid];
        for (int e = e0; e < e1; ++e) {
            const int key = key_of(e);
            const int val = (int)vin[e];
            const int dgt = (key >> shift) & 15;
            int pos = 0;
#pragma unroll
            for (int d = 0; d < 16; ++d)
                if (d == dgt) { pos = off[d]; off[d] = pos + 1; }
            if (last) { out_tok[pos] = key; out_rows[pos] = val; }
            else { kout[pos] = key; vout[pos] = (uint16_t)val; }
        }
        __syncthreads();
    }
}

// ---- embedding backward: dE[tok] = sum over the occurrences of tok of dX[row] (masked / scaled), tokens sorted ---------------------
// One workgroup per sorted position; the segment heads do the sums.  A head reads its occurrences eight at a time (row ids,
// gradient rows, mask words: each stage one batch of loads -- walked one by one, each occurrence was three dependent memory round
// trips, 2 us apiece for a token that occurs a few hundred times in a batch of natural text, with the whole launch waiting for that
// one workgroup).  LONG runs (ni = 512 only) are left to embed_scatter_long_kernel, whose 1024-thread workgroups split a run over 8
// thread groups: run [h, e) is long iff q0 + SC_LONG < e for q0 = the first multiple of SC_LONG at or behind h -- a rule both kernels
// can evaluate (this one knows h and e, the other one looks at toks[q - SC_LONG] and toks[q + SC_LONG]).  Sums are added in a fixed
// order everywhere: deterministic.
// (profiles/microbench/embed_scatter_zipf.py, us per scatter at the Yahoo shape, uniform ids / Zipf(1) ids whose most frequent token
//  occurs 628 times: one 128-thread group for everything 24.9 / 72.3; every position's workgroup with 2 groups 30.9 / 44.0, with 4
//  groups 43.5 / 35.7 -- the wider workgroups are paid for at every position; this two-kernel form 27.7 / 43.6, and in the Yahoo step
//  with Zipf ids (masked decoder scatter included) 72 -> 47 us per scatter on average)
constexpr int SC_LONG = 32;
constexpr int SC_GROUPS = 8;

// sum of the occurrences [q_lo, e) taken in batches of 8 with stride `stride` (one float4 column group, column k)
__device__ __forceinline__ float4 scatter_batches(const float* __restrict__ dX, const uint8_t* __restrict__ mask, float scale,
                                                  const int* __restrict__ rows, int q_lo, int e, int stride, int ni, int k, int B, int T) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int q0 = q_lo; q0 < e; q0 += stride) {
        int r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = rows[q0 + u < e ? q0 + u : e - 1];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(dX + (long)r[u] * ni + k);
        if (mask) {
            uint32_t mk[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int t = r[u] / B, b = r[u] % B;
                mk[u] = *reinterpret_cast<const uint32_t*>(mask + ((long)b * T + t) * ni + k);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                v[u].x = (mk[u] & 0xFFu) ? v[u].x * scale : 0.f;
                v[u].y = (mk[u] & 0xFF00u) ? v[u].y * scale : 0.f;
                v[u].z = (mk[u] & 0xFF0000u) ? v[u].z * scale : 0.f;
                v[u].w = (mk[u] & 0xFF000000u) ? v[u].w * scale : 0.f;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (q0 + u < e) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    return acc;
}

__global__ __launch_bounds__(128) void embed_scatter_kernel(const float* __restrict__ dX, const uint8_t* __restrict__ mask,
                                                            float scale, const int* __restrict__ rows,
                                                            const int* __restrict__ toks, int N, int B, int T,
                                                            float* __restrict__ dE, int ni, int pad_idx, int accumulate,
                                                            int vec, int V, int long_runs_elsewhere, float* __restrict__ sq, int sq_only) {
    const int p = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    if (V > 0 && !sq_only) {
        // complete form: every row of dE is written exactly once by this launch (pair) -- the rows of tokens that occur by their
        // segment heads, all others (and pad_idx) with zeros HERE, by the workgroup whose slice of the vocabulary they fall in
        // (balanced whatever the token distribution).  Whether a row occurs is a binary search of the sorted list: ONE search per
        // LANE, all rows of the slice at once -- 13 dependent loads per workgroup; walking the slice row by row it was 13 per row,
        // 3 rows per workgroup at the Yahoo shape, and most of the launch's 26-32 us.  Replaces a separate fill of the whole
        // table (41 MB at V = 20001).
        __shared__ int absent_row[64];
        __shared__ int s_cnt, s_win[128], s_pred[129];
        const int v0 = (int)((long)p * V / N), v1 = (int)((long)(p + 1) * V / N);
        // Round 6: the first 64 rows of the slice (all of it whenever V <= 64 N) are looked up through a WINDOW of the sorted list
        // that the whole workgroup fetches together -- the lower bound L of v0 by two counting steps (128 samples of the list, then
        // the <= N / 128 + 1 positions between two samples: one round trip each), then positions [L, L + 128) into LDS (a third);
        // a lane then searches its row in LDS.  A row the window does not reach (more than 128 occurrences inside the slice in front
        // of it) falls back to the search of the whole list.  3 dependent round trips instead of 13.
        int L = 0;
        if (v0 < v1) {
            // number of threads whose predicate holds, for a predicate that holds on a PREFIX of the threads
            auto prefix_count = [&](bool pred) -> int {
                s_pred[tid] = pred ? 1 : 0;
                if (tid == 0) s_pred[128] = 0;
                __syncthreads();
                if (pred && !s_pred[tid + 1]) s_cnt = tid + 1;
                if (tid == 0 && !pred) s_cnt = 0;
                __syncthreads();
                const int c = s_cnt;
                __syncthreads();
                return c;
            };
            // samples: s_i = the token at position ((i + 1) N) / 128 - 1 (no position for the leading i when N < 128: counted as below)
            const int sp = (int)(((long)(tid + 1) * N) / 128) - 1;
            const int sk = toks[sp >= 0 ? sp : 0];
            const int last_below = prefix_count(sp < 0 || sk < v0) - 1;      // the last sample below v0 (-1: none)
            // L lies behind that sample and not behind the next one (which is >= v0); the list is sorted: count what is below v0 there
            const int lo2 = (int)(((long)(last_below + 1) * N) / 128);
            const int hi2 = last_below + 1 < 128 ? (int)(((long)(last_below + 2) * N) / 128) : N;
            L = lo2;
            for (int c0 = lo2; c0 < hi2; c0 += 128) {
                const int q = c0 + tid;
                const int tk = toks[q < hi2 ? q : hi2 - 1];
                const int c = prefix_count(q < hi2 && tk < v0);
                L += c;
                if (c < 128) break;
            }
            const int wq = L + tid;
            s_win[tid] = wq < N ? toks[wq] : 0x7FFFFFFF;
            __syncthreads();
        }
        for (int vb = v0; vb < v1; vb += 64) {
            if (tid < 64) {
                const int v = vb + tid;
                bool present;
                if (vb == v0 && (s_win[127] >= v || L + 128 >= N)) {
                    int lo = 0, hi = 128;            // first window slot with a token >= v
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_win[mid] < v) lo = mid + 1; else hi = mid; }
                    present = lo < 128 && s_win[lo] == v;
                } else {
                    int lo = 0, hi = N;              // first position with toks[pos] >= v
                    while (lo < hi) { const int mid = (lo + hi) >> 1; if (toks[mid] < v) lo = mid + 1; else hi = mid; }
                    present = lo < N && toks[lo] == v;
                }
                absent_row[tid] = (v < v1 && !(present && v != pad_idx)) ? 1 : 0;
            }
            __syncthreads();
            const int nrow = v1 - vb < 64 ? v1 - vb : 64;
            for (int i = 0; i < nrow; ++i) {
                if (!absent_row[i]) continue;
                float* z = dE + (long)(vb + i) * ni;
                if (vec) for (int k = tid * 4; k < ni; k += 512) *reinterpret_cast<float4*>(z + k) = make_float4(0.f, 0.f, 0.f, 0.f);
                else for (int k = tid; k < ni; k += 128) z[k] = 0.f;
            }
            if (vb + 64 < v1) __syncthreads();
        }
    }
    // sq: the squares of the row this workgroup completes, one partial per wave in slots [2p, 2p + 2) (zeros from workgroups that
    // complete none); slots [2N, 2N + 2 ceil(N / SC_LONG)) belong to embed_scatter_long_kernel and are zeroed here when it does not run
    float ss = 0.f;
    const int tok = toks[p];
    bool work = !(p > 0 && toks[p - 1] == tok) && tok != pad_idx;      // a segment head of a real token
    if (work) {
        // end of the run [p, e): the next eight positions in one batch of loads (most tokens occur once or twice); a longer run is
        // finished by a binary search of the sorted list
        int e = p + 1;
        {
            int nx[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) nx[u] = toks[p + 1 + u < N ? p + 1 + u : N - 1];
            bool run = true;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                run = run && p + 1 + u < N && nx[u] == tok;
                if (run) e = p + 2 + u;
            }
            if (run && e < N) {                      // all eight equal: e = first position past p + 8 whose token differs
                int lo = e, hi = N;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (toks[mid] == tok) lo = mid + 1; else hi = mid; }
                e = lo;
            }
        }
        float* dst = dE + (long)tok * ni;
        if (vec) {
            if (!(long_runs_elsewhere && (p + SC_LONG - 1) / SC_LONG * SC_LONG + SC_LONG < e)) {      // (a long run is embed_scatter_long_kernel's)
                for (int k = tid * 4; k < ni; k += 512) {
                    float4 acc = scatter_batches(dX, mask, scale, rows, p, e, 8, ni, k, B, T);
                    float4* d4 = reinterpret_cast<float4*>(dst + k);
                    if (accumulate) { float4 o = *d4; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
                    ss += (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
                    if (!sq_only) *d4 = acc;
                }
            }
        } else {
            for (int k = tid; k < ni; k += 128) {
                float acc = 0.f;
                for (int q = p; q < e; ++q) {
                    const int r = rows[q];
                    float v = dX[(long)r * ni + k];
                    if (mask) {
                        const int t = r / B, b = r % B;
                        v = mask[((long)b * T + t) * ni + k] ? v * scale : 0.f;
                    }
                    acc += v;
                }
                if (accumulate) acc += dst[k];
                ss += acc * acc;
                if (!sq_only) dst[k] = acc;
            }
        }
    }
    if (sq) {
        ss = lv_wave_sum(ss);
        if ((tid & 63) == 0) {
            sq[2 * p + (tid >> 6)] = ss;
            if (!long_runs_elsewhere && p < (N + SC_LONG - 1) / SC_LONG) sq[2 * N + 2 * p + (tid >> 6)] = 0.f;
        }
    }
}

// the long runs (ni = 512, 16-byte aligned): workgroup j looks at position q = SC_LONG j and takes the run that contains it iff q is
// the run's first multiple of SC_LONG and the run reaches beyond q + SC_LONG; 8 groups of 128 threads take the run's batches of
// eight occurrences round-robin and their sums meet in LDS in group order
__global__ __launch_bounds__(128 * SC_GROUPS) void embed_scatter_long_kernel(const float* __restrict__ dX, const uint8_t* __restrict__ mask,
                                                                             float scale, const int* __restrict__ rows,
                                                                             const int* __restrict__ toks, int N, int B, int T,
                                                                             float* __restrict__ dE, int pad_idx, int accumulate,
                                                                             float* __restrict__ sq, int sq_only) {
    __shared__ __attribute__((aligned(16))) float part_sum[SC_GROUPS - 1][512];
    const int q = (int)blockIdx.x * SC_LONG;
    const int tid = (int)threadIdx.x & 127, grp = (int)threadIdx.x >> 7;
    // (sq: slots [2N + 2 blockIdx, + 2), the two waves of group 0; a workgroup that finds no run of its own writes zeros)
    float* sq_slot = sq && grp == 0 && (tid & 63) == 0 ? sq + 2 * N + 2 * (int)blockIdx.x + (tid >> 6) : nullptr;
    bool mine = q + SC_LONG < N;
    int tok = 0;
    if (mine) {
        tok = toks[q];
        mine = toks[q + SC_LONG] == tok && tok != pad_idx && !(q >= SC_LONG && toks[q - SC_LONG] == tok);      // (else an earlier workgroup's)
    }
    if (!mine) {
        if (sq_slot) *sq_slot = 0.f;
        return;
    }
    int lo = q - SC_LONG + 1 < 0 ? 0 : q - SC_LONG + 1, hi = q;     // head: first position in (q - SC_LONG, q] with this token
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (toks[mid] < tok) lo = mid + 1; else hi = mid; }
    const int h = lo;
    lo = q + SC_LONG + 1; hi = N;                                   // end: first position behind q + SC_LONG with another token
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (toks[mid] == tok) lo = mid + 1; else hi = mid; }
    const int e = lo;
    const int k = tid * 4;
    float4 acc = scatter_batches(dX, mask, scale, rows, h + 8 * grp, e, 8 * SC_GROUPS, 512, k, B, T);
    if (grp > 0) *reinterpret_cast<float4*>(&part_sum[grp - 1][k]) = acc;
    __syncthreads();
    if (grp != 0) return;
#pragma unroll
    for (int g = 0; g < SC_GROUPS - 1; ++g) {
        const float4 o = *reinterpret_cast<const float4*>(&part_sum[g][k]);
        acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w;
    }
    float4* d4 = reinterpret_cast<float4*>(dE + (long)tok * 512 + k);
    if (accumulate) { float4 o = *d4; acc.x += o.x; acc.y += o.y; acc.z += o.z; acc.w += o.w; }
    if (!sq_only) *d4 = acc;
    if (sq) {
        const float ss = lv_wave_sum((acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w));
        if (sq_slot) *sq_slot = ss;
    }
}

// Row-list exchange of an embedding gradient under data parallelism (of the V rows of the table's gradient at most
// B * T are non-zero on a rank): every rank contributes the sorted list of token ids that occur in its batch (ids_all[r][0 .. cap),
// ascending, padded with -1) and the gradient rows of those tokens (rows_all[r][i][:], f32 or bf16); the dense MEAN gradient is
// rebuilt by ONE workgroup per table row v: a binary search of v in every rank's list, the rows found are added in RANK ORDER
// (deterministic, the same bits on every rank), scaled, and written -- rows nobody touched are written as zeros, so dE is complete.
template <bool B16>
__global__ __launch_bounds__(128) void rows_merge_kernel(const int64_t* __restrict__ ids_all, const void* __restrict__ rows_all, int world,
                                                        int cap, int ni, int V, float scale, float* __restrict__ dE) {
    const int v = (int)blockIdx.x;
    const int tid = (int)threadIdx.x;
    __shared__ int pos[64];
    if (tid < world) {
        const int64_t* ids = ids_all + (long)tid * cap;
        int lo = 0, hi = cap;                                   // first index with ids[i] >= v among the valid (non-negative) prefix
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int64_t x = ids[mid];
            if (x >= 0 && x < v) lo = mid + 1; else hi = mid;
        }
        pos[tid] = (lo < cap && ids[lo] == v) ? lo : -1;
    }
    __syncthreads();
    for (int c = tid * 4; c < ni; c += 512) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        for (int r = 0; r < world; ++r) {
            const int q = pos[r];
            if (q < 0) continue;
            const long off = ((long)r * cap + q) * ni + c;
            if (B16) {
                const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(rows_all) + off);
                const uint32_t w4[4] = {u.x << 16, u.x & 0xFFFF0000u, u.y << 16, u.y & 0xFFFF0000u};      // bf16 = the high half of an f32
                float f4[4];
                memcpy(f4, w4, 16);
                a0 += f4[0]; a1 += f4[1]; a2 += f4[2]; a3 += f4[3];
            } else {
                const float4 f = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(rows_all) + off);
                a0 += f.x; a1 += f.y; a2 += f.z; a3 += f.w;
            }
        }
        *reinterpret_cast<float4*>(dE + (long)v * ni + c) = make_float4(a0 * scale, a1 * scale, a2 * scale, a3 * scale);
    }
}

}  // namespace

// dE [V][ni] = scale * sum over ranks r (in rank order) of the row of token v in rank r's list, zero where no rank has it.
// ids_all int64 [world][cap] (each list ascending, -1 padded), rows_all [world][cap][ni] f32 (b16 = 0) or bf16 (b16 = 1);
// ni % 4 == 0, world <= 64, 16-byte aligned buffers.  The data-parallel replacement for a dense all-reduce of the embedding
// gradient (reference: the gradient of nn.Embedding, enc_lstm.py:18, averaged over ranks as loss.mean() implies, text.py:382).
extern "C" int lv_rows_merge_f32(const int64_t* ids_all, const void* rows_all, int b16, int world, int cap, int ni, int V, float scale,
                                 float* dE, void* stream) {
    if (!ids_all || !rows_all || !dE) return LV_ERR_ARG;
    if (world <= 0 || world > 64 || cap <= 0 || ni <= 0 || ni % 4 != 0 || V <= 0) return LV_ERR_SHAPE;
    if ((((uintptr_t)rows_all) | ((uintptr_t)dE)) & 15) return LV_ERR_ALIGN;
    if (b16) LV_LAUNCH(rows_merge_kernel<true>, dim3((unsigned)V), dim3(128), 0, stream, ids_all, rows_all, world, cap, ni, V, scale, dE);
    else LV_LAUNCH(rows_merge_kernel<false>, dim3((unsigned)V), dim3(128), 0, stream, ids_all, rows_all, world, cap, ni, V, scale, dE);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// X[t*B + b][:] = emb[ids[b*ids_stride + t]][:] * (mask ? mask[b][t][:] * scale : 1)
extern "C" int lv_embed_gather_f32(const float* emb, const int64_t* ids, long ids_stride,
                                   const uint8_t* mask, float scale, float* X,
                                   int T, int B, int ni, int V, void* stream) {
    if (!emb || !ids || !X) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || ni <= 0 || V <= 0) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    const int vec = (ni % 4 == 0) && (((uintptr_t)emb | (uintptr_t)X) & 15) == 0;
    dim3 grid((unsigned)lv_cdiv((long)T * B, 4)), block(256);
    LV_LAUNCH(embed_gather_kernel, grid, block, 0, stream, emb, ids, ids_stride, mask, scale, X, T, B, ni, V, vec);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// Stable sort of rows r = t*B + b by token id.  out_rows/out_tok: [T*B] ints; tmp: 2*T*B ints.
extern "C" int lv_token_sort(const int64_t* ids, long ids_stride, int T, int B, int V,
                             int* out_rows, int* out_tok, int* tmp, void* stream) {
    if (!ids || !out_rows || !out_tok || !tmp) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || V <= 0) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    if ((long)T * B <= SORT_LDS_MAX)
        LV_LAUNCH(token_sort_lds_kernel, dim3(1), dim3(SORT_THREADS), 0, stream, ids, ids_stride, T, B, V, out_rows, out_tok);
    else
        LV_LAUNCH(token_sort_kernel, dim3(1), dim3(SORT_THREADS), 0, stream, ids, ids_stride, T, B, V, out_rows, out_tok, tmp);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// dE[tok][:] (=|+=) sum over rows r with token tok of dX[r][:] * (mask ? mask[b][t][:]*scale : 1), in ascending r.
// Rows of dE for tokens that do not occur are left untouched; pad_idx (or -1) is skipped entirely.
extern "C" int lv_embed_scatter_f32(const float* dX, const uint8_t* mask, float scale,
                                    const int* sorted_rows, const int* sorted_tok, int T, int B,
                                    float* dE, int ni, int pad_idx, int accumulate, void* stream) {
    if (!dX || !sorted_rows || !sorted_tok || !dE) return LV_ERR_ARG;
    if (T < 0 || B <= 0 || ni <= 0) return LV_ERR_SHAPE;
    if (T == 0) return LV_OK;
    const int vec = (ni % 4 == 0) && (((uintptr_t)dX | (uintptr_t)dE) & 15) == 0 && (((uintptr_t)mask) & 3) == 0;      // (mask bytes read 4 at a time)
    const int N = T * B;
    const int two = vec && ni == 512 && N > 2 * SC_LONG;
    LV_LAUNCH(embed_scatter_kernel, dim3((unsigned)N), dim3(128), 0, stream, dX, mask, scale, sorted_rows, sorted_tok,
              N, B, T, dE, ni, pad_idx, accumulate, vec, 0, two, (float*)nullptr, 0);
    if (two)
        LV_LAUNCH(embed_scatter_long_kernel, dim3((unsigned)lv_cdiv(N, SC_LONG)), dim3(128 * SC_GROUPS), 0, stream, dX, mask, scale, sorted_rows,
                  sorted_tok, N, B, T, dE, pad_idx, accumulate, (float*)nullptr, 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

// The same sums, and dE COMPLETE: rows [0, V) of tokens that do not occur (and pad_idx) are written as zeros by the same launch,
// so the caller needs no fill of the table in front of it.  Every token id must lie in [0, V).
static int embed_scatter_full(const float* dX, const uint8_t* mask, float scale, const int* sorted_rows, const int* sorted_tok, int T, int B,
                              float* dE, int ni, int V, int pad_idx, float* sq, int sq_only, void* stream) {
    if (!dX || !sorted_rows || !sorted_tok || !dE) return LV_ERR_ARG;
    if (T <= 0 || B <= 0 || ni <= 0 || V <= 0) return LV_ERR_SHAPE;
    const int vec = (ni % 4 == 0) && (((uintptr_t)dX | (uintptr_t)dE) & 15) == 0 && (((uintptr_t)mask) & 3) == 0;      // (mask bytes read 4 at a time)
    const int N = T * B;
    const int two = vec && ni == 512 && N > 2 * SC_LONG;
    LV_LAUNCH(embed_scatter_kernel, dim3((unsigned)N), dim3(128), 0, stream, dX, mask, scale, sorted_rows, sorted_tok,
              N, B, T, dE, ni, pad_idx, 0, vec, V, two, sq, sq ? sq_only : 0);
    if (two)
        LV_LAUNCH(embed_scatter_long_kernel, dim3((unsigned)lv_cdiv(N, SC_LONG)), dim3(128 * SC_GROUPS), 0, stream, dX, mask, scale, sorted_rows,
                  sorted_tok, N, B, T, dE, pad_idx, 0, sq, sq ? sq_only : 0);
    LV_CHECK_LAUNCH();
    return LV_OK;
}

extern "C" int lv_embed_scatter_full_f32(const float* dX, const uint8_t* mask, float scale,
                                         const int* sorted_rows, const int* sorted_tok, int T, int B,
                                         float* dE, int ni, int V, int pad_idx, void* stream) {
    return embed_scatter_full(dX, mask, scale, sorted_rows, sorted_tok, T, B, dE, ni, V, pad_idx, nullptr, 0, stream);
}

// The complete scatter and, in the same pass, the sum of squares of the table gradient it completes: sq[0 .. parts) receives one
// partial per wave (fixed slots: deterministic; rows no token touches are zero and contribute nothing), parts =
// lv_embed_scatter_sumsq_parts(T, B).  sq_only != 0: dE is NOT written (a gradient needed for the norm of clip_grad_norm_ alone).
extern "C" int lv_embed_scatter_sumsq_parts(int T, int B) {
    const long N = (long)T * B;
    return N <= 0 ? 0 : (int)(2 * N + 2 * ((N + SC_LONG - 1) / SC_LONG));
}

extern "C" int lv_embed_scatter_full_sumsq_f32(const float* dX, const uint8_t* mask, float scale,
                                               const int* sorted_rows, const int* sorted_tok, int T, int B,
                                               float* dE, int ni, int V, int pad_idx, float* sq, int sq_only, void* stream) {
    if (!sq) return LV_ERR_ARG;
    return embed_scatter_full(dX, mask, scale, sorted_rows, sorted_tok, T, B, dE, ni, V, pad_idx, sq, sq_only, stream);
}
