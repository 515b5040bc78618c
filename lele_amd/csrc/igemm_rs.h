// igemm_rs.h -- register-stationary i8 GEMMs for the quantised linears of a batch of utterances (included by quant.hip).
//
// Same arithmetic as igemm_kernel (exact i32 products on v_mfma_i32_32x32x32_i8, IgemmEpi's f32 epilogue:
// /root/reference/src/kernels/avx/quantization.rs:225-417, 1396-1428), another schedule.  The K = 512 / 2048 products of a
// SenseVoice-shaped layer over 5472 rows are short and wide; a tiled kernel spends its life in barriers and global -> VGPR -> LDS
// round trips (round 2: 8-10 % MFMA busy, 17-20 us per launch for 2-6 us of bytes).  Here:
//
//   * both operands live in HBM in MFMA FRAGMENT ORDER: block (tile, k-step) = 1 KiB, lane l's 16 bytes at offset 16 l hold
//     row / column 32 tile + (l & 31), k = 32 step + 16 (l >> 5) + [0, 16) -- every operand load of a wave is ONE fully
//     coalesced 1 KiB request (weights: wpack_frag_kernel, once; activations: qrows_frag_kernel, which quantises into that
//     order; the hidden layer of a feed-forward block: the epilogue of the pass that produces it, whose 32 x 32 result tile IS
//     one k-step block of the next product);
//   * K = 512 (igemm_rs_kernel): a consumer wave keeps the weight fragments of ITS 32 columns for the whole K extent in 64 VGPRs
//     and multiplies the 32-row activation tiles that loader waves put into an LDS ring -- FOUR waves that read the f32 rows, derive
//     the slices' quantisation parameters and quantise on the way (FQ: K = 512 exactly, 16-byte aligned rows; nothing is launched in
//     front of the kernel), or two waves streaming ready-made i8 fragments with direct-to-LDS loads (K padded to 512, odd
//     alignments); one s_barrier per tile; the eight consumers of a workgroup (eight column tiles) share every activation tile;
//   * the weights are the MFMA's FIRST operand, so a lane owns ONE result row and 4 x 4 consecutive columns of it: 16-byte
//     stores, one set of row terms per lane (they travel with the tile through the ring), the column terms in registers;
//   * K = 2048 (igemm_rs_ks4_kernel, the second feed-forward product): the four waves of a workgroup split K, A fragments
//     double-buffered in registers, partial tiles meet in LDS (one barrier per tile), row sums come from v_dot4 on the fragments
//     the wave loads anyway;
//   * K = 512, at most 512 columns (igemm_as_kernel, the projection behind the attention): the other way round -- a workgroup keeps
//     ONE 32-row tile, quantised once into LDS, and its waves bring the weight fragments of all (or 4 / 8 of the) column tiles.
#pragma once

namespace {

struct RsArgs {
    const int8_t* af;  // activations, fragment-major [nrt][ks][1024]
    const int8_t* wf;  // weights, fragment-major [nct][ks][1024]
    unsigned rows;
    int n;
    int nrt, nct;  // 32-row tiles, 32-column tiles
    int ncb, nrr;  // K = 512: column blocks of 8 tiles and row ranges (grid = ncb * nrr); K split: grid = nct * nrr
    int8_t* hid;   // EM 2: the result as fragment-major i8 [nrt][n / 32][1024]
    const float* x;  // FQ kernels: the f32 activation [rows][512] itself (af is unused): the loaders quantise it into the ring
    // FQ kernels derive the slices' parameters themselves (no parameter kernel in front):
    const float* partial;  // {min, max} pairs of the activation, [slices][nblk]
    int nblk;
    unsigned* hpart;       // fused feed-forward block: [grid][4] bits of the hidden layer's maxima per workgroup (its first four slices) --
                           // EM 1 writes them (plain stores: nothing to clear beforehand), EM 2 reduces the ones that concern its rows
    unsigned* sync;        // EM 3 (both passes in one launch): [64] launch counters, one a row range, never cleared (a launch adds ncb to each);
                           // behind them [grid][4] 8-byte words {launch tag, bits of the maximum}: what hpart is to EM 1 / 2
    unsigned* deverr;      // EM 3: the context's sticky error word (a wait that ran out of time: LELE_DEVERR_FFN_SYNC)
#ifdef LELE_HIP_LAB
    long long* dbg;  // lab build: [grid][9][32] wall-clock stamps (100 MHz) of every wave (wave 8 = the loader), or NULL
    int ablate;      // lab build (results wrong): 1 no products, 2 no epilogue, 4 no fragment reads (products on stale registers)
#endif
};
#ifdef LELE_HIP_LAB
#define RS_STAMP(who) do { if (g.dbg && lane == 0 && nstamp < 32 && wave < 9) g.dbg[((size_t)blockIdx.x * 9 + wave) * 32 + nstamp++] = (long long)wall_clock64(); } while (0)
#else
#define RS_STAMP(who) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------ rows -> fragment-major i8
// f32 rows [rows][k] -> q - 128 as i8 in fragment order [nrt][kp / 32][1024] (+ exact i32 row sums), quantised with the range
// of the row's batch slice exactly as qrows_kernel<0> does (SIMD body rint(fma(x, 1/scale, zp)), scalar tail round(x*inv + zp)).
// A wave owns 256 chunks of 16 elements = 8 rows (kp = 512) or 2 rows (kp = 2048); all its row data is requested before the
// range partials of its slice(s) are reduced.  Rows beyond `rows` of the last tile and bytes beyond k are written as 0.
template <int CPR /* 16-element chunks per row: kp / 16 */, int IT /* trips of 64 chunks per wave */>
__global__ __launch_bounds__(256) void qrows_frag_kernel(const float* __restrict__ x, unsigned rows, int k, int m,
                                                         QParams* __restrict__ prm, int8_t* __restrict__ af,
                                                         int* __restrict__ row_sums, const float* __restrict__ partial, int nblk,
                                                         unsigned* __restrict__ zero_slice) {
    static_assert((CPR == 32 && IT == 1) || (CPR == 128 && IT == 4), "kp = 512: two rows per wave; kp = 2048: two rows per wave");
    constexpr int KS = CPR / 2;
    const int lane = threadIdx.x & 63;
    const unsigned gw = blockIdx.x * 4u + (threadIdx.x >> 6);
    const unsigned c0 = gw * (64u * IT);                 // first chunk of this wave (row-major over [rows padded][CPR])
    const unsigned row_first = c0 / CPR, row_last = (c0 + 64u * IT - 1u) / CPR;
    const bool vec_ok = (k & 3) == 0 && (((uintptr_t)x & 15) == 0);
    const int simd_k = k & ~7;
    float4 v[IT][4];
    unsigned rowi[IT], ci[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const unsigned ch = c0 + 64u * i + lane;
        rowi[i] = ch / CPR;
        ci[i] = ch % CPR;
        const unsigned r = rowi[i] < rows ? rowi[i] : rows - 1u;
        const float* src = x + (size_t)r * k + 16u * ci[i];
        if (vec_ok && (int)(16u * ci[i]) + 16 <= k) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = *reinterpret_cast<const float4*>(src + 4 * e);
        } else {
            float t[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int kk = (int)(16u * ci[i]) + e;
                t[e] = x[(size_t)r * k + (kk < k ? kk : k - 1)];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[i][e] = make_float4(t[4 * e], t[4 * e + 1], t[4 * e + 2], t[4 * e + 3]);
        }
    }
    if (row_first >= rows) {  // a wave of padding rows only
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const size_t blk = (size_t)(rowi[i] >> 5) * KS + (ci[i] >> 1);
            *reinterpret_cast<v4i*>(af + (blk * 64 + (rowi[i] & 31u) + 32u * (ci[i] & 1u)) * 16) = v4i{0, 0, 0, 0};
        }
        return;
    }
    const unsigned mu = (unsigned)m;
    const unsigned last_valid = row_last < rows ? row_last : rows - 1u;
    const unsigned s_lo = row_first / mu, s_hi = last_valid / mu;
    unsigned pk[IT][4];
    int sum[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        sum[i] = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) pk[i][e] = 0u;
    }
    for (unsigned s = s_lo; s <= s_hi; ++s) {  // wave-uniform: the slices this wave's rows belong to (one or two in practice)
        QParams q;
        if (partial) {
            float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
            const float2* pp = reinterpret_cast<const float2*>(partial) + (size_t)s * nblk;
            for (int i0 = 0; i0 < nblk; i0 += 256) {  // 4 pairs per lane in flight per trip (clamped: a repeated pair changes nothing)
                float2 w[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int idx = i0 + lane + 64 * u;
                    w[u] = pp[idx < nblk ? idx : nblk - 1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    mn = w[u].x < mn ? w[u].x : mn;
                    mx = w[u].y > mx ? w[u].y : mx;
                }
            }
            mn = wave_allreduce64(mn, [](float cur, float a) { return a < cur ? a : cur; });
            mx = wave_allreduce64(mx, [](float cur, float a) { return a > cur ? a : cur; });
            q = make_qparams(mn, mx);
            // the wave that holds the slice's first row publishes its parameters (every row lies in exactly one wave)
            if (lane == 0 && s * mu >= row_first && s * mu <= row_last) {
                prm[s] = q;
                if (zero_slice) zero_slice[s] = 0u;
            }
        } else {
            q = prm[s];
            if (zero_slice && lane == 0 && s * mu >= row_first && s * mu <= row_last) zero_slice[s] = 0u;
        }
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            if (rowi[i] >= rows || rowi[i] / mu != s) continue;
            const int kb = (int)(16u * ci[i]);
            if (kb + 16 <= simd_k) {  // the whole chunk inside the SIMD body
                unsigned us = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    unsigned p = 0u;
                    p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[i][e].x, q.inv_scale, q.zp), 0, p);
                    p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[i][e].y, q.inv_scale, q.zp), 1, p);
                    p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[i][e].z, q.inv_scale, q.zp), 2, p);
                    p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(v[i][e].w, q.inv_scale, q.zp), 3, p);
                    us = __builtin_amdgcn_sad_u8(p, 0u, us);
                    pk[i][e] = p ^ 0x80808080u;
                }
                sum[i] = (int)us - 128 * 16;
            } else {
                const float xv[16] = {v[i][0].x, v[i][0].y, v[i][0].z, v[i][0].w, v[i][1].x, v[i][1].y, v[i][1].z, v[i][1].w,
                                      v[i][2].x, v[i][2].y, v[i][2].z, v[i][2].w, v[i][3].x, v[i][3].y, v[i][3].z, v[i][3].w};
                int sacc = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int kk = kb + e;
                    int val = 0;
                    if (kk < k) {
                        val = (int)quant_one(xv[e], q, kk < simd_k) - 128;
                        sacc += val;
                    }
                    pk[i][e >> 2] |= (unsigned)(val & 0xff) << (8 * (e & 3));
                }
                sum[i] = sacc;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const size_t blk = (size_t)(rowi[i] >> 5) * KS + (ci[i] >> 1);
        *reinterpret_cast<v4i*>(af + (blk * 64 + (rowi[i] & 31u) + 32u * (ci[i] & 1u)) * 16) =
            v4i{(int)pk[i][0], (int)pk[i][1], (int)pk[i][2], (int)pk[i][3]};
    }
    if (!row_sums) return;
    auto addi = [](float a, float b) { return __int_as_float(__float_as_int(a) + __float_as_int(b)); };
    if constexpr (CPR == 32) {  // a row = the 32 chunks of one half wave of one trip
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int tot = __float_as_int(group_allreduce32(__int_as_float(sum[i]), addi));
            if ((lane & 31) == 0 && rowi[i] < rows) row_sums[rowi[i]] = tot;
        }
    } else {  // a row = the 128 chunks of two consecutive trips
#pragma unroll
        for (int i = 0; i < IT; i += 2) {
            const int tot = wave_sum_i32(sum[i] + sum[i + 1]);
            if (lane == 0 && rowi[i] < rows) row_sums[rowi[i]] = tot;
        }
    }
}

// ------------------------------------------------------------------------------------------ shared pieces
__device__ __forceinline__ unsigned rs_logical_block() {  // workgroup b runs on XCD b % 8: give every XCD a contiguous range
    const unsigned G = gridDim.x, b = blockIdx.x, xcd = b & 7u, base = G >> 3, rem = G & 7u;
    return xcd * base + (xcd < rem ? xcd : rem) + (b >> 3);
}

// {scale, zp, 1 / scale, (int) zp} of slice `sl` from the producer's {min, max} pairs -- what qparams_kernel / qrows_kernel<0> compute,
// by one wave (min / max are order-independent: the same parameters whoever reduces)
__device__ __forceinline__ QParams slice_params(const float* __restrict__ partial, int nblk, unsigned sl, int lane) {
    float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
    const float2* pp = reinterpret_cast<const float2*>(partial) + (size_t)sl * nblk;
    for (int i0 = 0; i0 < nblk; i0 += 256) {  // 4 pairs per lane in flight per trip (clamped: a repeated pair changes nothing)
        float2 w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = i0 + lane + 64 * j;
            w[j] = pp[idx < nblk ? idx : nblk - 1];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            mn = w[j].x < mn ? w[j].x : mn;
            mx = w[j].y > mx ? w[j].y : mx;
        }
    }
    mn = wave_allreduce64(mn, [](float cur, float a) { return a < cur ? a : cur; });
    mx = wave_allreduce64(mx, [](float cur, float a) { return a > cur ? a : cur; });
    return make_qparams(mn, mx);
}

struct RsRow {  // what the epilogue needs to know about a lane's row of the tile in flight
    int rowsum;
    float scale;
    int zp_i;
    unsigned slice;
    unsigned smax;  // EM 2: bits of the hidden layer's maximum over the row's slice
    float q2_scale, q2_zp, q2_inv;  // EM 2 behind quantising loaders: the hidden layer's parameters, ready-made
};

// f32 value of one result element: IgemmEpi::value24 with the row terms in registers
__device__ __forceinline__ float rs_value(int acc, int rterm, int ca, int colsum, float dsws, float bias, bool has_ws, bool has_bias,
                                          int relu) {
    const int total = acc + rterm + __mul24(ca, colsum);
    float vf = (float)total;  // _mm256_cvtepi32_ps
    if (has_ws) vf = vf * dsws;
    if (has_bias) vf = vf + bias;
    if (relu) vf = relu0(vf);
    return vf;
}

// ------------------------------------------------------------------------------------------ K = 512: weights in registers, activations through an LDS ring
// EM 0: f32 result (+ up to two residual operands, + optional {min, max} per (row tile, column tile) for a single-slice consumer)
// EM 1: only the per-slice maximum of the ReLU result (atomic max on the bits), EM 2: the result quantised with that range, written
//       as the fragment-major i8 operand of the next product (see lele_hip_fused_ffn_quantized)
//
// Ten waves (twelve with quantising loaders, FQ): eight CONSUMERS, each with the weight fragments of its own 32 columns for the whole K extent in 64 VGPRs, and two
// LOADERS that do nothing but direct-to-LDS loads (global_load_lds_dwordx4: a fragment block is 1 KiB of consecutive lanes, which
// is exactly the lane-linear image that instruction writes) of the workgroup's 32-row activation tiles into a ring of RS_NS slots.
// What the CU's load path delivers is the bound of these products (measured: ~21 TB/s chip-wide = 40 B/clk/CU whether the lines
// are shared or private, tools/l2bw.hip), so every activation byte crosses it ONCE per workgroup instead of once per wave,
// and it is requested up to four tiles ahead of its use.  The loader's vmcnt counter sees only its own 16 loads per tile, so
// "tile i has landed" is an exact counted wait (the tiles issued after it stay in flight); one s_barrier per tile hands tile i to
// the consumers and the slot of tile i - 1 back to the loader.  Consumers never touch a counter by hand: their loads (weights
// once, row terms, residuals) and stores are the compiler's.
constexpr int RS_NS = 5, RS_AHEAD = 3, RS_TILE = 16 * 1024, RS_SLOT = RS_TILE + 4 * 256;  // a slot: the tile's 16 fragment blocks + its row terms
constexpr int RS_MAXSL = 16;                                          // FQ: slices a workgroup's row range may touch (launch_rs checks)
constexpr int RS_LDS = RS_NS * RS_SLOT + 3 * 8 * 32 * 4 + 16 + 2 * RS_MAXSL * 16;  // ring + column strips + EM 1 maxima + FQ parameter tables
// EM 3: every tile of the workgroup keeps its slot (nothing is quantised twice): at most RS_NS_BOTH tiles a workgroup
constexpr int RS_NS_BOTH = 8;
constexpr int RS_LDS_BOTH = RS_NS_BOTH * RS_SLOT + 3 * 8 * 32 * 4 + 16 + 2 * RS_MAXSL * 16;

// one direct-to-LDS load of 16 bytes per lane: LDS address = lds_dst (wave-uniform byte address) + 16 * lane.  Inline asm: the
// compiler's own counter bookkeeping must not see it (it would drain the ring at every barrier)
__device__ __forceinline__ void rs_dma16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
__device__ __forceinline__ void rs_dma4(const void* gsrc, unsigned lds_dst) {  // 4 bytes per lane: LDS address = lds_dst + 4 * lane
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_dst)
                 : "memory");
}
template <int N>
__device__ __forceinline__ void rs_wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory");
}
__device__ __forceinline__ void rs_barrier() { asm volatile("s_barrier" ::: "memory"); }

// FQ ("fused quantise"): FOUR loader waves read the f32 ROWS, reduce the parameters of the slices their workgroup's rows belong to from
// the producer's {min, max} pairs (slice_params: what qparams_kernel / qrows_kernel<0> compute), quantise exactly as qrows_frag_kernel
// does -- rint(fma(x, 1/scale, zp)) saturated to u8 (one v_cvt_pk_u8_f32), minus 128 -- and write the codes into the ring in fragment
// order themselves.  The separate quantising pass (10.7 us per call on a configs[3] shard: an 11 MB
// read and a 2.8 MB write between two kernels of 14 us) and the fragment-major copy of the activation in HBM disappear; each row tile
// is quantised once per column block (6-8 times over the grid), out of L2.  A loader lane owns 4 consecutive k of one row per load:
// 16 rows x 64 bytes per wave instruction on the global side, 64 different LDS banks on the ds_write_b32 side.
//
// EM 3 (FQ only): the feed-forward block's range pass AND its quantise pass in ONE launch.  The two passes are the same products twice;
// as two kernels each pays the launch, the weights-in-registers prologue and the loaders' quantisation of every row tile (the
// knock-outs of DESIGN.md 3.3: 11 of a pass's 17 us).  Here every tile of the workgroup keeps its ring slot (RS_NS_BOTH slots: nothing
// is overwritten), pass 1 runs as EM 1 does, the workgroup publishes its four maxima -- each as ONE 8-byte word {launch tag, bits of
// the maximum} written at device scope -- and its loaders read the words of the workgroups whose row range shares a slice with
// theirs (all their ncb column blocks) until they carry this launch's tag: the wait and the fetch are one load.  They reduce the
// hidden layer's parameters as EM 2's loaders do, and the consumers run the products again straight out of LDS -- no loader, no
// barrier between tiles -- into EM 2's epilogue.  Same maxima, same parameters, same codes: bit-identical to the two launches.
// The tag is the launch number: every launch adds exactly ncb to each row range's counter (g.sync[rr], never cleared), so the value
// a workgroup's own increment returns (asked for at entry, used after the first pass) divided by ncb is the same for every workgroup.
// Waiting on other workgroups needs them to be RESIDENT: the grid is at most one workgroup per CU (rs_grid), workgroups are
// dispatched in index order and a workgroup only waits for row ranges next to its own, so the launch is safe alone on the device;
// the host side (ffn_impl) takes this route only on lane 0 of the only context of the device with no side lane in flight, and a
// wait that lasts longer than RS_SYNC_LIMIT sets LELE_DEVERR_FFN_SYNC and goes on (the next sync reports it) instead of hanging.
constexpr long long RS_SYNC_LIMIT = 20 * 1000 * 100;  // 20 ms of the 100 MHz wall clock
template <int EM, int NRES, bool RELU, bool FQ = false>
__global__ __launch_bounds__(768) void igemm_rs_kernel(RsArgs g, IgemmEpi epi) {
    constexpr int KS = 16;
    constexpr bool BOTH = EM == 3;
    constexpr bool RANGE = EM == 1 || BOTH;   // the workgroup gathers the maxima of its slices
    constexpr int NS = BOTH ? RS_NS_BOTH : RS_NS;
    static_assert(!BOTH || FQ, "EM 3 is a form of the quantising-loader kernel");
    extern __shared__ __attribute__((aligned(16))) char rs_lds[];
    char* const ring = rs_lds;
    int* const s_colsum = reinterpret_cast<int*>(rs_lds + NS * RS_SLOT);  // [8][32] each
    float* const s_ws = reinterpret_cast<float*>(s_colsum + 256);
    float* const s_bias = s_ws + 256;
    unsigned* const s_mx = reinterpret_cast<unsigned*>(s_bias + 256);  // [4]
    float4* const s_prm = reinterpret_cast<float4*>(s_mx + 4);           // FQ: [RS_MAXSL] {scale, zp, 1 / scale, (int) zp} of the input's slices
    float4* const s_q2 = s_prm + RS_MAXSL;                                // FQ, EM 2: [RS_MAXSL] {scale, zp, 1 / scale, bits of the maximum} of the hidden layer's
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hv = lane >> 5, l31 = lane & 31;
    const unsigned L = rs_logical_block();
    const int rr = (int)(L / (unsigned)g.ncb), cb = (int)(L - (unsigned)rr * (unsigned)g.ncb);
    const int t0 = (int)((unsigned)rr * (unsigned)g.nrt / (unsigned)g.nrr), t1 = (int)((unsigned)(rr + 1) * (unsigned)g.nrt / (unsigned)g.nrr);
    if (t0 >= t1) return;  // uniform over the workgroup
    const int nt = t1 - t0;
    if (RANGE) {  // the workgroup's slice maxima meet here before they go to memory
        if (threadIdx.x < 4) s_mx[threadIdx.x] = 0u;
        __syncthreads();
    }
    if (FQ && wave >= 8) {
        // ---------------------------------------------------------------- the four loaders, quantising
        // wave 8 + j owns rows [16 u, 16 u + 16) of every tile (u = j / 2) and k in [256 half, 256 half + 256) (half = j % 2): 4 lanes a
        // row, 16 float4s a lane and tile = 64 registers, so TWO tiles fit in flight.
        const int half = (wave - 8) & 1, u = (wave - 8) >> 1, q4 = lane & 3, rsub = lane >> 2;
        const unsigned rows = g.rows, mu = (unsigned)epi.m;
        const bool single = rows == mu;
        const unsigned row_end = (unsigned)t1 * 32u < rows ? (unsigned)t1 * 32u : rows;
        const unsigned s_lo = single ? 0u : ((unsigned)t0 * 32u) / mu, s_hi = single ? 0u : (row_end - 1u) / mu;  // at most RS_MAXSL slices
        auto load = [&](int i, float4 (&dst)[16]) {
            unsigned row = (unsigned)(t0 + i) * 32u + 16u * (unsigned)u + (unsigned)rsub;
            row = row < rows ? row : rows - 1u;  // rows beyond the end repeat the last one: multiplied, never stored
            const float* src = g.x + (size_t)row * 512u + 256u * (unsigned)half + 4u * (unsigned)q4;
#pragma unroll
            for (int c = 0; c < 16; ++c) dst[c] = *reinterpret_cast<const float4*>(src + 16 * c);
        };
        auto put = [&](int i, const float4 (&src)[16]) {
            char* const slot = ring + (i % NS) * RS_SLOT;
            unsigned row = (unsigned)(t0 + i) * 32u + 16u * (unsigned)u + (unsigned)rsub;
            row = row < rows ? row : rows - 1u;
            const unsigned rel = single ? 0u : row / mu - s_lo;
            const float4 q = s_prm[rel];  // {scale, zp, 1 / scale, (int) zp}
            const float inv_scale = q.z, zp = q.y;
            unsigned us = 0u;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                unsigned p = 0u;
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(src[c].x, inv_scale, zp), 0, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(src[c].y, inv_scale, zp), 1, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(src[c].z, inv_scale, zp), 2, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(src[c].w, inv_scale, zp), 3, p);
                us = __builtin_amdgcn_sad_u8(p, 0u, us);
                // k-step 8 half + c / 2 of the tile; inside its 1 KiB block lane (row, (c & 1)) holds k = 16 (c & 1) + [0, 16)
                *reinterpret_cast<unsigned*>(slot + (8 * half + (c >> 1)) * 1024 + ((16 * u + rsub) + 32 * (c & 1)) * 16 + 4 * q4) = p ^ 0x80808080u;
            }
            us += (unsigned)__shfl_xor((int)us, 1);  // the row's four lanes: its 256 codes of this half
            us += (unsigned)__shfl_xor((int)us, 2);
            if (q4 == 0) {
                const int r = 16 * u + rsub;
                reinterpret_cast<int*>(slot + RS_TILE)[32 * half + r] = (int)us - 128 * 256;  // the two halves' sums side by side
                if (half == 0) {
                    reinterpret_cast<float*>(slot + RS_TILE + 256)[r] = q.x;
                    reinterpret_cast<int*>(slot + RS_TILE + 512)[r] = __float_as_int(q.w);
                    if (EM == 2) {  // the hidden layer's parameters of the row's slice, ready-made for the eight consumers
                        const float4 q2 = s_q2[rel];
                        reinterpret_cast<float*>(slot + RS_TILE + 768)[r] = q2.w;
                        reinterpret_cast<float*>(slot + RS_TILE + 256)[32 + r] = q2.x;
                        reinterpret_cast<float*>(slot + RS_TILE + 512)[32 + r] = q2.y;
                        reinterpret_cast<float*>(slot + RS_TILE + 768)[32 + r] = q2.z;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the codes are in LDS before anybody is told so
        };
        // Even tiles travel through RA, odd ones through RB (two named sets: nothing is copied, the compiler's counted waits stay exact).
        // A tile's rows are requested two barriers before it is handed over; barrier number i (counted from 0) hands over tile i.
        // (Loads are unconditional -- beyond the range the last tile is read again and dropped: a load under a branch makes its
        // registers a merge of two values, and the copy the compiler then needs waits for the load on the spot.)
        float4 RA[16], RB[16];
        load(0, RA);
        // The parameters of the slices the workgroup's rows belong to, while the first rows travel: loader j takes slices s_lo + j,
        // s_lo + j + 4, ... -- the input's from the producer's {min, max} pairs; EM 2: the hidden layer's from the maxima the range pass's
        // workgroups left (every workgroup whose row range meets the slice, all its column blocks).
        // the hidden layer's parameters of slice sl from the maxima the range pass's workgroups left: every workgroup whose row range
        // meets the slice, all its column blocks
        const unsigned nrt_u = (unsigned)g.nrt, nrr_u = (unsigned)g.nrr, ncb_u = (unsigned)g.ncb;
        auto range_of = [&](unsigned tile) {  // the row range whose tiles [rr nrt / nrr, (rr + 1) nrt / nrr) hold `tile`
            unsigned r = tile * nrr_u / nrt_u;
            while (r + 1u < nrr_u && (r + 1u) * nrt_u / nrr_u <= tile) ++r;
            while (r > 0u && r * nrt_u / nrr_u > tile) --r;
            return r;
        };
        auto ranges_of_slice = [&](unsigned sl, unsigned& ra, unsigned& rb) {
            const unsigned r_first = sl * mu, r_last = (sl + 1u) * mu < rows ? (sl + 1u) * mu - 1u : rows - 1u;
            ra = single ? 0u : range_of(r_first >> 5);
            rb = single ? nrr_u - 1u : range_of(r_last >> 5);
        };
        unsigned tag = 0u;  // EM 3: this launch's number + 1 (set before the first use)
        auto hidden_params = [&](unsigned sl) {
            unsigned ra, rb;
            ranges_of_slice(sl, ra, rb);
            const unsigned total = (rb - ra + 1u) * ncb_u;
            float mxh = 0.0f;  // ReLU results: >= 0, never NaN -- their bits order like the values
            for (unsigned e0 = 0; e0 < total; e0 += 64u) {
                const unsigned e = e0 + (unsigned)lane;
                const unsigned ec = e < total ? e : total - 1u;
                const unsigned r = ra + ec / ncb_u, c = ec - (ec / ncb_u) * ncb_u;
                const unsigned sb = single ? 0u : ((r * nrt_u / nrr_u) * 32u) / mu;  // that workgroup's first slice
                const size_t at = (size_t)(r * ncb_u + c) * 4u + (sl - sb);
                if constexpr (BOTH) {
                    // written by workgroups of THIS launch, possibly on another XCD: {launch tag, maximum} as ONE 8-byte word, read at
                    // device scope (past this XCD's L2) until the tag is this launch's -- the wait and the fetch are the same load
                    const unsigned long long* hp = reinterpret_cast<const unsigned long long*>(g.sync + 64) + at;
                    const long long started = (long long)wall_clock64();
                    unsigned long long v;
                    for (;;) {
                        v = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (__all((unsigned)(v >> 32) == tag)) break;
                        if ((long long)wall_clock64() - started > RS_SYNC_LIMIT) {
                            if (lane == 0) atomicOr(g.deverr, 2u);  // LELE_DEVERR_FFN_SYNC (the word lives in host memory)
                            break;
                        }
                        __builtin_amdgcn_s_sleep(4);
                    }
                    mxh = fmaxf(mxh, __uint_as_float((unsigned)v));
                } else {
                    mxh = fmaxf(mxh, __uint_as_float(g.hpart[at]));
                }
            }
            mxh = wave_allreduce64(mxh, [](float cur, float a) { return fmaxf(cur, a); });
            const QParams q2 = make_qparams(0.0f, mxh);
            if (lane == 0) s_q2[sl - s_lo] = make_float4(q2.scale, q2.zp, q2.inv_scale, mxh);
        };
        // EM 3: the launch number comes from the row range's counter, which every launch advances by ncb: asked for now, needed after the
        // first pass
        unsigned ticket = 0u;
        if (BOTH && wave == 8 && lane == 0) ticket = __hip_atomic_fetch_add(g.sync + rr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (unsigned sl = s_lo + (unsigned)(wave - 8); sl <= s_hi; sl += 4u) {
            const QParams qs = slice_params(g.partial, g.nblk, sl, lane);
            if (lane == 0) s_prm[sl - s_lo] = make_float4(qs.scale, qs.zp, qs.inv_scale, __int_as_float(qs.zp_i));
            if (EM == 2) hidden_params(sl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        rs_barrier();  // the tables are complete (the consumers take this barrier too)
        for (int i = 0; i < nt; i += 2) {
            load(i + 1 < nt ? i + 1 : nt - 1, RB);
            if (i > 0) rs_barrier();  // tile i - 1
            put(i, RA);
            load(i + 2 < nt ? i + 2 : nt - 1, RA);
            rs_barrier();  // tile i
            if (i + 1 < nt) put(i + 1, RB);
        }
        if ((nt & 1) == 0) rs_barrier();  // an odd tile was the last one
        if (RANGE) __syncthreads();  // the consumers' maxima are in s_mx
        if (BOTH) {
            // Publish the workgroup's four maxima, each with the launch's tag in the upper half of ONE 8-byte word, at device scope (the
            // stores write through this XCD's L2; nothing is flushed or invalidated for them -- a release / acquire pair at device scope
            // writes back and invalidates the whole L2: measured, the launch 9 us SLOWER than the two it replaces).  The tag every
            // workgroup of a row range computes is the same (the counter's old value / ncb), and so is the neighbours' (all counters
            // advance together), so the loaders below wait for exactly the words this launch writes.
            ticket = (unsigned)__shfl((int)ticket, 0);
            if (wave == 8) {
                tag = ticket / ncb_u + 1u;
                if (lane < 4)   // slices s_lo .. s_lo + 3 of this workgroup
                    __hip_atomic_store(reinterpret_cast<unsigned long long*>(g.sync + 64) + (size_t)L * 4u + (unsigned)lane,
                                       ((unsigned long long)tag << 32) | s_mx[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                reinterpret_cast<unsigned*>(s_q2 + RS_MAXSL - 1)[3] = tag;  // the other loaders take the tag from here
            }
            __syncthreads();  // (the consumers run the first tile's products of the second pass meanwhile)
            tag = reinterpret_cast<const unsigned*>(s_q2 + RS_MAXSL - 1)[3];
            for (unsigned sl = s_lo + (unsigned)(wave - 8); sl <= s_hi; sl += 4u) hidden_params(sl);
            __syncthreads();  // the hidden layer's parameters are in s_q2: the consumers run the second pass on their own
        }
        return;
    }
    if (!FQ && wave >= 8) {
        // ---------------------------------------------------------------- the two loaders of ready-made fragments
        // One wave issues a 1 KiB direct-to-LDS load every ~45 ns (measured; MI355X_MICROARCH.md's "ldsdma-fill" row: ~25 GB/s per
        // CU and loader wave), i.e. 0.8 us per tile -- as long as a consumer's whole tile.  So two waves split a tile's blocks
        // (wave 8: k-steps 0-7 and the row terms; wave 9: k-steps 8-15), and a tile is ISSUED before the loader waits for an
        // older one and goes to the barrier: with five slots the slot of tile i + 3 was last read before barrier i - 1.
        const int half = wave - 8;
        const unsigned ring_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ring;
        constexpr int NTERM = EM == 2 ? 4 : 3, HS = KS / 2;
        const unsigned rows = g.rows, mu = (unsigned)epi.m, nslices = rows / mu;
        auto issue = [&](int i) {
            const char* src = reinterpret_cast<const char*>(g.af) + (size_t)(t0 + i) * (KS * 1024) + half * (HS * 1024) + lane * 16;
            const unsigned dst = __builtin_amdgcn_readfirstlane(ring_base + (unsigned)(i % NS) * RS_SLOT);
#pragma unroll
            for (int s = 0; s < HS; ++s) rs_dma16(src + s * 1024, dst + (half * HS + s) * 1024);
            if (half == 0) {
                // the tile's row terms (both half waves fetch the same 32 rows): row sum, dynamic scale, zero point (, slice maximum)
                unsigned row = (unsigned)(t0 + i) * 32u + (unsigned)(lane & 31);
                row = row < rows ? row : rows - 1u;
                const unsigned sl = rows == mu ? 0u : row / mu;
                rs_dma4(epi.row_sums + row, dst + RS_TILE);
                rs_dma4(&epi.prm[sl].scale, dst + RS_TILE + 256);
                rs_dma4(&epi.prm[sl].zp_i, dst + RS_TILE + 512);
                if (EM == 2) rs_dma4(epi.slice_max + (sl < nslices ? sl : nslices - 1u), dst + RS_TILE + 768);
            }
        };
        // "tile i has landed": the (up to RS_AHEAD) tiles issued after it may stay in flight -- an exact counted wait, this
        // wave's vmcnt counter sees nothing but its own PER loads per tile
        auto wait_for = [&](int younger, auto per_c) {
            constexpr int PER = decltype(per_c)::value;
            static_assert(RS_AHEAD == 3 && 3 * PER <= 63, "the wave's vmcnt counter has six bits");
            if (younger >= 3) rs_wait_vm<3 * PER>();
            else if (younger == 2) rs_wait_vm<2 * PER>();
            else if (younger == 1) rs_wait_vm<PER>();
            else rs_wait_vm<0>();
        };
        int nstamp = 0;
        RS_STAMP(1);
        const int pre = nt < RS_AHEAD ? nt : RS_AHEAD;
        for (int i = 0; i < pre; ++i) issue(i);
        RS_STAMP(1);
        for (int i = 0; i < nt; ++i) {
            if (i + RS_AHEAD < nt) issue(i + RS_AHEAD);
            const int younger = nt - 1 - i < RS_AHEAD ? nt - 1 - i : RS_AHEAD;
            if (half == 0) wait_for(younger, std::integral_constant<int, HS + NTERM>());
            else wait_for(younger, std::integral_constant<int, HS>());
            RS_STAMP(1);   // [2 + 2 i] this wave's part of tile i landed
            rs_barrier();  // tile i is in LDS for everybody
            RS_STAMP(1);   // [3 + 2 i] the consumers arrived
        }
        if (EM == 1) __syncthreads();
        return;
    }
    // -------------------------------------------------------------------- a consumer
    const int ct = cb * 8 + wave;
    if (ct >= g.nct) {  // a partial last column block: the wave only keeps the barriers company
        if (FQ) rs_barrier();
        for (int i = 0; i < nt; ++i) rs_barrier();
        if (RANGE) __syncthreads();
        if (BOTH) {
            __syncthreads();
            __syncthreads();
        }
        return;
    }
    int nstamp = 0;
    RS_STAMP(0);   // [0] entry
#ifdef LELE_HIP_LAB
    const long long cyc0 = clock64();
#endif
    const int n = g.n;
    const unsigned rows = g.rows, mu = (unsigned)epi.m;
    const bool single = rows == mu;
    v4i bf[KS];
    {
        const v4i* wp = reinterpret_cast<const v4i*>(g.wf) + (size_t)ct * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) bf[s] = wp[s * 64];
    }
    // the column terms of the lane's 16 columns (ct * 32 + 8 g + 4 hv + e) stay in registers: 48 of them, and no LDS round trip
    // in front of every column group of every tile
    int colsum[16];
    float wsc[16], biasc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        int c = ct * 32 + 8 * (q >> 2) + 4 * hv + (q & 3);
        c = c < n ? c : n - 1;  // clamped: loads stay unconditional, out-of-range columns are never stored
        colsum[q] = epi.col_sums[c];
        wsc[q] = epi.wscale_len <= 1 ? epi.wscale[0] : epi.wscale[c];
        biasc[q] = epi.bias ? epi.bias[c] : -0.0f;  // x + (-0.0) == x for every x, the sign of zero included
    }
    const int cbz = 128 - epi.zp_b;
    // results leave through buffer stores: a lane that has nothing to store points beyond the buffer and the hardware drops
    // it -- no branch in the epilogue, so the next tile's products and this tile's epilogue are ONE basic block to the scheduler
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc(EM == 0 ? (void*)epi.out : (void*)epi.q_prm,
                                                            0, EM == 0 ? (int)(rows * (unsigned)n * 4u) : (int)((rows / mu) * 16u), 0x00020000);
    (void)out_rsrc;
    // EM 1: running maxima of the (up to four) slices the workgroup's row range touches; anything further away goes out at once
    const unsigned sbase = single ? 0u : ((unsigned)t0 * 32u) / mu;
    float mx[4] = {0.0f, 0.0f, 0.0f, 0.0f};

    // the products of tile i out of its ring slot, and the row terms the loader put behind it (dynamic quantisation is the only
    // caller: epi.prm is never NULL here)
    auto products = [&](int i, v16i& acc, RsRow& r) {
        const char* const slot = ring + (i % NS) * RS_SLOT;
        const v4i* ap = reinterpret_cast<const v4i*>(slot) + lane;
        r.rowsum = reinterpret_cast<const int*>(slot + RS_TILE)[l31];
        if (FQ) r.rowsum += reinterpret_cast<const int*>(slot + RS_TILE)[32 + l31];  // the quantising loaders leave one sum per half of k
        r.scale = reinterpret_cast<const float*>(slot + RS_TILE + 256)[l31];
        r.zp_i = reinterpret_cast<const int*>(slot + RS_TILE + 512)[l31];
        r.smax = EM == 2 ? reinterpret_cast<const unsigned*>(slot + RS_TILE + 768)[l31] : 0u;
        if (EM == 2 && FQ) {
            r.q2_scale = reinterpret_cast<const float*>(slot + RS_TILE + 256)[32 + l31];
            r.q2_zp = reinterpret_cast<const float*>(slot + RS_TILE + 512)[32 + l31];
            r.q2_inv = reinterpret_cast<const float*>(slot + RS_TILE + 768)[32 + l31];
        }
        // the accumulators START from the terms that do not depend on the products -- (128 - zp_b) rowsum + K (128 - zp_a)(128 - zp_b)
        // + (128 - zp_a) colsum: i32 additions in another order, the same total, and 16 vector instructions a tile fewer than adding them
        // behind the matrix core
        {
            const int ca = 128 - r.zp_i;
            const int rterm = cbz * r.rowsum + epi.k * ca * cbz;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = __mul24(ca, colsum[q]) + rterm;
        }
#ifdef LELE_HIP_LAB
        if (g.ablate & 1) {
        } else if (g.ablate & 4) {
#pragma unroll
            for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], bf[(s + 1) & 15], acc, 0, 0, 0);
        } else
#endif
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], ap[s * 64], acc, 0, 0, 0);
            if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most four fragments read ahead: 16 registers, not 64
        }
    };
    auto epilogue = [&](auto mode_c, int i, const v16i& acc, const RsRow& r) {
        constexpr int EMODE = decltype(mode_c)::value;  // the epilogue's own mode: EM 3 runs the one of EM 1, then the one of EM 2
#ifdef LELE_HIP_LAB
        if (g.ablate & 2) {
            if (acc[0] == 0x12345 && r.rowsum == 77) epi.out[0] = 1.0f;
            return;
        }
#endif
        const int t = t0 + i;
        const unsigned row = (unsigned)t * 32u + (unsigned)l31;
        const bool rok = row < rows;
        const unsigned rowc = rok ? row : rows - 1u;
        const unsigned obase = rowc * (unsigned)n + (unsigned)(ct * 32 + 4 * hv);  // rows * n < 2^30 (rs_fits)
        const unsigned slice = (EMODE == 0 || single) ? 0u : rowc / mu;
        float4 res1[NRES > 0 ? 4 : 1], res2[NRES > 1 ? 4 : 1];
        if (NRES > 0) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const int c = ct * 32 + 4 * hv + 8 * gq;
                const unsigned at = c < n ? obase + 8 * gq : rowc * (unsigned)n + (unsigned)(n - 4);
                res1[gq] = *reinterpret_cast<const float4*>(epi.res1 + at);
                if (NRES > 1) res2[gq] = *reinterpret_cast<const float4*>(epi.res2 + at);
            }
        }
        const float ds = r.scale;
        QParams q2;
        if (EMODE == 2) {
            if (BOTH) {  // the second pass of EM 3: the loaders left the parameters per slice, not per row of a slot
                const float4 t = s_q2[slice - sbase];
                q2 = QParams{t.x, t.y, t.z, (int)t.y};
            } else if (FQ) q2 = QParams{r.q2_scale, r.q2_zp, r.q2_inv, (int)r.q2_zp};
            else q2 = make_qparams(0.0f, __uint_as_float(r.smax));
        }
        float4 o[4];
        unsigned d[4];
        // IgemmEpi::value24 (the weight scale is mandatory on this route): _mm256_cvtepi32_ps, x (dynamic scale x weight scale), + bias,
        // ReLU -- each its own rounding.  Scalar f32 instructions on purpose (and -fno-slp-vectorize for this file): v_pk_mul_f32 /
        // v_pk_add_f32 beside the matrix core cost more than the two instructions they replace (MI355X_MICROARCH.md, measured here:
        // the packed epilogue made the range and quantise passes 1.3 us slower)
        auto val = [&](int total, float dsws, float b) {
#ifdef LELE_HIP_LAB
            if (g.ablate & 16) return __int_as_float(total);
#endif
            float vf = (float)total;
            vf = vf * dsws;
            vf = vf + b;
            if (RELU) vf = relu0(vf);
            return vf;
        };
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            o[gq].x = val(acc[4 * gq + 0], ds * wsc[4 * gq + 0], biasc[4 * gq + 0]);
            o[gq].y = val(acc[4 * gq + 1], ds * wsc[4 * gq + 1], biasc[4 * gq + 1]);
            o[gq].z = val(acc[4 * gq + 2], ds * wsc[4 * gq + 2], biasc[4 * gq + 2]);
            o[gq].w = val(acc[4 * gq + 3], ds * wsc[4 * gq + 3], biasc[4 * gq + 3]);
            if (EMODE == 0) {
                if (NRES > 0) {
                    o[gq].x = o[gq].x + res1[gq].x, o[gq].y = o[gq].y + res1[gq].y, o[gq].z = o[gq].z + res1[gq].z, o[gq].w = o[gq].w + res1[gq].w;
                    if (NRES > 1)
                        o[gq].x = o[gq].x + res2[gq].x, o[gq].y = o[gq].y + res2[gq].y, o[gq].z = o[gq].z + res2[gq].z, o[gq].w = o[gq].w + res2[gq].w;
                }
                const bool cok = ct * 32 + 4 * hv + 8 * gq < n;  // n % 4 == 0: a group of four columns is whole or absent
                const v4u bits = {__float_as_uint(o[gq].x), __float_as_uint(o[gq].y), __float_as_uint(o[gq].z), __float_as_uint(o[gq].w)};
#ifdef LELE_HIP_LAB
                if (g.ablate & 8) {
                    if (bits[0] == 0x12345u) epi.out[1] = 2.0f;
                } else
#endif
                __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, rok && cok ? (obase + 8u * gq) * 4u : 0xffffffffu, 0, 0);
            } else if (EMODE == 2) {
                unsigned p = 0u;
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(o[gq].x, q2.inv_scale, q2.zp), 0, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(o[gq].y, q2.inv_scale, q2.zp), 1, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(o[gq].z, q2.inv_scale, q2.zp), 2, p);
                p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(o[gq].w, q2.inv_scale, q2.zp), 3, p);
                d[gq] = p ^ 0x80808080u;
            }
        }
        if (EMODE == 2) {
            // a lane holds columns {0-3, 8-11, 16-19, 24-27} + 4 hv of its row; two half-wave exchanges turn that into the 16
            // consecutive columns 16 hv + [0, 16): exactly lane (row, hv)'s bytes of the consumer's fragment block (row tile t, k-step ct)
            const auto r02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
            const auto r13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
            const v4i chunk = {(int)r02[0], (int)r02[1], (int)r13[0], (int)r13[1]};
            *(reinterpret_cast<v4i*>(g.hid) + ((size_t)t * g.nct + ct) * 64 + lane) = chunk;
            const v4u qb = {__float_as_uint(q2.scale), __float_as_uint(q2.zp), __float_as_uint(q2.inv_scale), (unsigned)q2.zp_i};
            __builtin_amdgcn_raw_buffer_store_b128(qb, out_rsrc, ct == 0 && hv == 0 && rok && row == slice * mu ? slice * 16u : 0xffffffffu, 0, 0);
        }
        if (EMODE == 1) {  // ReLU results: >= 0, NaN became 0; n % 32 == 0 on the two-pass route: every column of the tile exists
            float lmax = 0.0f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) lmax = fmaxf(lmax, fmaxf(fmaxf(o[gq].x, o[gq].y), fmaxf(o[gq].z, o[gq].w)));
            if (!rok) lmax = 0.0f;
            const unsigned rel = slice - sbase;
            mx[0] = rel == 0u ? fmaxf(mx[0], lmax) : mx[0];
            mx[1] = rel == 1u ? fmaxf(mx[1], lmax) : mx[1];
            mx[2] = rel == 2u ? fmaxf(mx[2], lmax) : mx[2];
            mx[3] = rel == 3u ? fmaxf(mx[3], lmax) : mx[3];
            if (!FQ && rel > 3u && lmax > 0.0f) atomicMax(&epi.slice_max[slice], __float_as_uint(lmax));  // (FQ: launch_rs admits four slices a workgroup)
        }
        if (EMODE == 0 && epi.blockstat) {  // one {min, max} pair per (row tile, column tile): LeleBuf::rowstat kind 1
            float smn = 3.40282347e+38f, smx = -3.40282347e+38f;
#pragma unroll
            for (int gq = 0; gq < 4; ++gq)
                if (rok && ct * 32 + 4 * hv + 8 * gq < n) {
                    smn = fminf(smn, fminf(fminf(o[gq].x, o[gq].y), fminf(o[gq].z, o[gq].w)));
                    smx = fmaxf(smx, fmaxf(fmaxf(o[gq].x, o[gq].y), fmaxf(o[gq].z, o[gq].w)));
                }
            smn = wave_allreduce64(smn, [](float cur, float x) { return x < cur ? x : cur; });
            smx = wave_allreduce64(smx, [](float cur, float x) { return x > cur ? x : cur; });
            if (lane == 0) {
                float* so = epi.blockstat + ((size_t)t * g.nct + ct) * 2;
                so[0] = smn;
                so[1] = smx;
            }
        }
    };
    // One tile per barrier interval: products, then the epilogue.  (Measured alternatives, all within 5 % of this and fatter in
    // registers: the next tile's products woven into this tile's epilogue with sched_group_barrier; the two waves of a SIMD in
    // opposite phase.  The epilogue of a lone wave is dependency-bound, not issue-bound, so neither arrangement buys overlap.)
    v16i acc;
    RsRow r;
    if (FQ) rs_barrier();  // the loaders' parameter tables (nothing of the consumers' depends on them: it only keeps the count)
    for (int i = 0; i < nt; ++i) {
        rs_barrier();  // tile i has landed
        RS_STAMP(0);
        products(i, acc, r);
        if (i == 0) {
#pragma unroll
            for (int s = 0; s < KS; ++s) asm volatile("" : "+v"(bf[s]));  // the weights are in: later tiles wait for nothing of the prologue
        }
        RS_STAMP(0);
        epilogue(std::integral_constant<int, BOTH ? 1 : EM>(), i, acc, r);
    }
    RS_STAMP(0);
#ifdef LELE_HIP_LAB
    if (g.dbg && lane == 0) g.dbg[((size_t)blockIdx.x * 9 + wave) * 32 + 31] = clock64() - cyc0;  // shader cycles entry -> end
#endif
    if (RANGE) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float v = wave_allreduce64(mx[j], [](float cur, float x) { return fmaxf(cur, x); });
            if (lane == 0 && v > 0.0f) atomicMax(&s_mx[j], __float_as_uint(v));  // non-negative floats order like their bits
        }
        __syncthreads();
        if (BOTH) {
            // loader wave 8 publishes the maxima and waits for the neighbours; then the loaders reduce the hidden layer's parameters
            RS_STAMP(0);  // the workgroup's maxima are complete
            __syncthreads();
            products(0, acc, r);  // the second pass: every tile is still in its slot; the first products need no parameter
            RS_STAMP(0);
            __syncthreads();
            RS_STAMP(0);  // the hidden layer's parameters are in LDS
            for (int i = 0; i < nt; ++i) {
                if (i > 0) products(i, acc, r);
                RS_STAMP(0);
                epilogue(std::integral_constant<int, 2>(), i, acc, r);
                RS_STAMP(0);
            }
        } else if (FQ) {
            if (threadIdx.x < 4) g.hpart[(size_t)L * 4u + threadIdx.x] = s_mx[threadIdx.x];  // slices sbase .. sbase + 3 of this workgroup
        } else if (threadIdx.x < 4 && s_mx[threadIdx.x]) {
            atomicMax(&epi.slice_max[sbase + threadIdx.x], s_mx[threadIdx.x]);
        }
    }
}

// ------------------------------------------------------------------------------------------ K = 2048: four waves split K
// One workgroup = one 32-column tile over a range of row tiles; wave w holds the weight fragments of k-steps [16 w, 16 w + 16)
// and multiplies the same quarter of every activation tile; the four partial tiles (and the partial row sums: v_dot4 with ones
// over the fragments in registers) meet in a double-buffered LDS exchange, ONE barrier per tile; wave w then finishes column
// group w of the tile (4 columns per lane, one 16-byte store).  The activation's row sums never exist in HBM.
template <int NRES>
__global__ __launch_bounds__(256, 2) void igemm_rs_ks4_kernel(RsArgs g, IgemmEpi epi) {
    constexpr int KS = 64, KW = 16;
    __shared__ v4i s_x[2][4][4][64];  // [buffer][wave][column group][lane]
    __shared__ int s_rs[2][4][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hv = lane >> 5, l31 = lane & 31;
    const unsigned L = rs_logical_block();
    const int rr = (int)(L / (unsigned)g.nct), ct = (int)(L - (unsigned)rr * (unsigned)g.nct);
    const int t0 = (int)((int64_t)rr * g.nrt / g.nrr), t1 = (int)((int64_t)(rr + 1) * g.nrt / g.nrr);
    if (t0 >= t1) return;  // uniform over the workgroup
    const int n = g.n;
    const unsigned rows = g.rows, mu = (unsigned)epi.m;
    const bool has_ws = epi.wscale != nullptr, has_bias = epi.bias != nullptr, single = rows == mu;

    v4i bf[KW];
    {
        const v4i* wp = reinterpret_cast<const v4i*>(g.wf) + ((size_t)ct * KS + wave * KW) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KW; ++s) bf[s] = wp[s * 64];
    }
    // the four columns this lane finishes: ct * 32 + 8 wave + 4 hv + e
    const int col = ct * 32 + 8 * wave + 4 * hv;
    const bool cok = col < n;  // n % 4 == 0
    int colsum[4];
    float ws[4], bias[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int c = col + e < n ? col + e : n - 1;
        colsum[e] = epi.col_sums[c];
        ws[e] = has_ws ? (epi.wscale_len <= 1 ? epi.wscale[0] : epi.wscale[c]) : 1.0f;
        bias[e] = has_bias ? epi.bias[c] : 0.0f;
    }
    const int cbz = 128 - epi.zp_b;
    struct Row {
        float scale;
        int zp_i;
    };
    auto load_tile = [&](v4i (&a)[KW], Row& r, int t) {
        const v4i* ap = reinterpret_cast<const v4i*>(g.af) + ((size_t)t * KS + wave * KW) * 64 + lane;
#pragma unroll
        for (int s = 0; s < KW; ++s) a[s] = ap[s * 64];
        unsigned row = (unsigned)t * 32u + (unsigned)l31;
        row = row < rows ? row : rows - 1u;
        r.scale = 1.0f;
        r.zp_i = epi.zp_a_fixed;
        if (epi.prm) {
            const unsigned sl = single ? 0u : row / mu;
            r.scale = epi.prm[sl].scale;
            r.zp_i = epi.prm[sl].zp_i;
        }
    };
    auto compute = [&](const v4i (&a)[KW], const Row& r, int t, int buf) {
        const unsigned row = (unsigned)t * 32u + (unsigned)l31;
        const bool rok = row < rows;
        const size_t at = (size_t)(rok ? row : rows - 1u) * (unsigned)n + (unsigned)(cok ? col : n - 4);
        float4 res1, res2;
        if (NRES > 0) res1 = *reinterpret_cast<const float4*>(epi.res1 + at);
        if (NRES > 1) res2 = *reinterpret_cast<const float4*>(epi.res2 + at);
        v16i acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0;
        int rsp = 0;
#pragma unroll
        for (int s = 0; s < KW; ++s) {
            acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf[s], a[s], acc, 0, 0, 0);
            rsp = __builtin_amdgcn_sdot4(a[s][0], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[s][1], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[s][2], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[s][3], 0x01010101, rsp, false);
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) s_x[buf][wave][gq][lane] = v4i{acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]};
        s_rs[buf][wave][lane] = rsp;
        __syncthreads();  // the only barrier of a tile: buffer `buf` is rewritten two tiles later, after the next tile's barrier
        v4i tot = s_x[buf][0][wave][lane];
        int rowsum = s_rs[buf][0][l31] + s_rs[buf][0][l31 + 32];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const v4i p = s_x[buf][w][wave][lane];
            tot[0] += p[0], tot[1] += p[1], tot[2] += p[2], tot[3] += p[3];
            rowsum += s_rs[buf][w][l31] + s_rs[buf][w][l31 + 32];
        }
        const int ca = 128 - r.zp_i;
        const int rterm = cbz * rowsum + epi.k * ca * cbz;
        float4 o;
        o.x = rs_value(tot[0], rterm, ca, colsum[0], r.scale * ws[0], bias[0], has_ws, has_bias, epi.relu);
        o.y = rs_value(tot[1], rterm, ca, colsum[1], r.scale * ws[1], bias[1], has_ws, has_bias, epi.relu);
        o.z = rs_value(tot[2], rterm, ca, colsum[2], r.scale * ws[2], bias[2], has_ws, has_bias, epi.relu);
        o.w = rs_value(tot[3], rterm, ca, colsum[3], r.scale * ws[3], bias[3], has_ws, has_bias, epi.relu);
        if (NRES > 0) o.x = o.x + res1.x, o.y = o.y + res1.y, o.z = o.z + res1.z, o.w = o.w + res1.w;
        if (NRES > 1) o.x = o.x + res2.x, o.y = o.y + res2.y, o.z = o.z + res2.z, o.w = o.w + res2.w;
        if (rok && cok) *reinterpret_cast<float4*>(epi.out + at) = o;
    };
    v4i a0[KW], a1[KW];
    Row r0, r1;
    load_tile(a0, r0, t0);
    int t = t0;
    while (t + 1 < t1) {  // straight-line pairs, as in igemm_rs_kernel
        load_tile(a1, r1, t + 1);
        compute(a0, r0, t, 0);
        load_tile(a0, r0, t + 2 < t1 ? t + 2 : t1 - 1);
        compute(a1, r1, t + 1, 1);
        t += 2;
    }
    if (t < t1) compute(a0, r0, t, 0);
}

// ------------------------------------------------------------------------------------------ K = 512, N <= 512: activations stationary
// The projection behind the attention (5472 x 512 x 512 on a configs[3] shard) has too few column tiles for weights-in-registers (a
// workgroup would multiply one or two row tiles with 128 KB of weights) and, as a tiled kernel, paid a separate quantising pass over
// the rows (9 us) in front of 12 us of GEMM.  Here a workgroup owns ONE 32-row tile and ALL columns: its eight waves request the
// weight fragments of their two column tiles (fragment order: 32 fully coalesced 1 KiB loads a wave, 128 registers), quantise the tile's
// f32 rows meanwhile -- once, into LDS, with the slices' parameters reduced from the producer's {min, max} partials on the spot --
// and then run straight through: 16 fragment reads, 32 products, epilogue.  No loop, one barrier, no intermediate in HBM.
struct AsArgs {
    const float* x;        // f32 rows [rows][512]
    const int8_t* wf;      // weights, fragment-major [nct][16][1024]
    unsigned rows;
    int n, nct;
    const float* partial;  // [slices][nblk] {min, max} pairs
    int nblk;
    QParams* prm;          // published per slice (the wave that holds the slice's first row)
    int tpw;               // !TWO: column tiles per workgroup (8 or 4: grid.y = ceil(nct / tpw)); wave w multiplies tile blockIdx.y tpw + w
    // LN (n == 512, TWO): the LayerNorm that follows the projection, in the same launch (lele_hip_fused_quantized_linear_residual_ln)
    const float* ln_g = nullptr;    // [512] scale
    const float* ln_b = nullptr;    // [512] bias
    float ln_eps = 0.0f;
    float* ln_out = nullptr;        // [rows][512]
    float* ln_rowstat = nullptr;    // NULL or [rows][2]: {min, max} of every normalised row (LeleBuf::rowstat kind 0)
    // FS (with LN): res1 is not read but COMPUTED here -- the FSMN memory block depthwise_conv1d_tlc(fs_x, fs_w, fs_b, pl, pr, add_input)
    // over the workgroup's 32 rows (conv.hip dwconv1d_tlc_kernel: the same FMA chain per output, so the same bits)
    const float* fs_x = nullptr;    // [rows][fs_pitch], already offset to the first of the 512 channels; epi.m time steps per utterance
    int fs_pitch = 0;
    const float* fs_w = nullptr;    // [512][FS]
    const float* fs_b = nullptr;    // NULL or [512]
    // (padding FS / 2 on either side: the output is as long as the input)
};
constexpr int AS_LN_PITCH = 516;    // words: eight rows x four words of a ds_write_b128 lane group fall on 32 different banks
constexpr int AS_LN_LDS = 32 * AS_LN_PITCH * 4;
constexpr int as_fs_lds(int fs) { return (32 + fs - 1) * AS_LN_PITCH * 4; }

// TWO = false (few row tiles: one utterance): a workgroup takes only `tpw` column tiles, one a wave, and the grid's second dimension
// the rest -- the row tile is quantised once per column group (out of L2), in exchange for four times the workgroups.
// The LayerNorm phase of the activation-stationary kernels: the workgroup's 32 x 512 result sits in LDS (s_ln, pitch AS_LN_PITCH);
// half-wave (wave, hv) normalises tile rows 2 wave + hv and 16 + 2 wave + hv with the operations of layer_norm_reg_kernel<16> in
// their order (eltwise.hip; avx/norm.rs:10-133: the same bits), stores them and leaves {min, max} per row for the next quantiser.
__device__ __forceinline__ void as_ln_rows(const float* s_ln, const float* s_lng, const float* s_lnb, unsigned t, int wave, int hv, int l31,
                                           unsigned rows, float eps, float* ln_out, float* ln_rowstat) {
    constexpr int NT = 16;
    const float inv_n = 1.0f / 512.0f;
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + hv + 16 * rr;
        const unsigned grow = t * 32u + (unsigned)r;
        float v[NT];
#pragma unroll
        for (int c = 0; c < NT; ++c) v[c] = s_ln[r * AS_LN_PITCH + 32 * c + l31];
        float sum, sumsq;
        row_sums_reg<NT, true, true>(v, 512, l31, &sum, &sumsq);
        const float mean = sum * inv_n;
        const float var = sumsq * inv_n - mean * mean;
        const float inv_std = 1.0f / sqrtf(var + eps);
        float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
        float* orow = ln_out + (size_t)(grow < rows ? grow : rows - 1u) * 512u;
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            const float tv = (v[c] - mean) * inv_std;
            const float o = fmaf_(tv, s_lng[32 * c + l31], s_lnb[32 * c + l31]);  // 512 % 8 == 0: every element is in the 8-wide body
            if (grow < rows) orow[32 * c + l31] = o;
            mn = o < mn ? o : mn;
            mx = o > mx ? o : mx;
        }
        if (ln_rowstat) {
            mn = group_allreduce32(mn, [](float cur, float a) { return a < cur ? a : cur; });
            mx = group_allreduce32(mx, [](float cur, float a) { return a > cur ? a : cur; });
            if (l31 == 0 && grow < rows) {
                ln_rowstat[2 * (size_t)grow] = mn;
                ln_rowstat[2 * (size_t)grow + 1] = mx;
            }
        }
    }
}

// LN = true: the workgroup holds whole rows of the result (32 x 512), so the LayerNorm that reads it runs here as well -- the tile goes
// through LDS once (a lane of the epilogue owns 32 columns of ONE row, the normalisation wants a row spread over 32 lanes), then every
// half-wave normalises two rows with the operations of layer_norm_reg_kernel<16> in their order (eltwise.hip; avx/norm.rs:10-133:
// the same bits), writes them and leaves their {min, max} for the quantiser of the next linear.  One launch, one read of the sum less.
// FS = the FSMN kernel width (0: none).  The first residual of a SAN-M layer's out projection is the memory block
// conv_k(v) + v of the SAME rows' v (k = 11 time steps around each row, zero beyond the utterance): a workgroup fetches the 32 + k - 1
// rows of v it needs straight into the tile region (direct-to-LDS loads, no registers), thread c slides a k-register window down
// column c and overwrites row r with its result in place (it is the only reader of its column), and the epilogue takes res1 from
// there -- the 11 MB result of the separate kernel, its launch and its re-read are gone.
template <int NRES, bool RELU, bool TWO = true, bool LN = false, int FS = 0>
__global__ __launch_bounds__(512) void igemm_as_kernel(AsArgs g, IgemmEpi epi) {
    static_assert(!LN || TWO, "the LayerNorm needs whole rows in one workgroup");
    static_assert(FS == 0 || (LN && NRES >= 1), "the in-kernel FSMN block is res1 of the LayerNorm form");
    constexpr int KS = 16;
    extern __shared__ __attribute__((aligned(16))) float s_ln[];  // LN: [32][AS_LN_PITCH]
    __shared__ float s_lng[LN ? 512 : 1], s_lnb[LN ? 512 : 1];
    __shared__ __attribute__((aligned(16))) char s_tile[KS * 1024];
    __shared__ int s_rowsum[4][32];
    __shared__ float s_scale[32];
    __shared__ int s_zp[32];
    __shared__ __attribute__((aligned(16))) int s_colsum[512];
    __shared__ __attribute__((aligned(16))) float s_ws[512];
    __shared__ __attribute__((aligned(16))) float s_bias[512];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hv = lane >> 5, l31 = lane & 31;
    const int t = blockIdx.x, n = g.n;
    const unsigned rows = g.rows, mu = (unsigned)epi.m;
    // Everything below is requested in the order it is needed (a wave's loads complete in order as far as its counter can tell): the
    // range partials of the first two slices, the rows, the column terms -- and only then the 32 KiB of weights, which travel while
    // the parameters are reduced and the rows quantised.
    // ---- 1. the rows: wave (u, kq) takes rows [16 u, 16 u + 16) and k in [128 kq, 128 kq + 128): 4 lanes a row, 8 chunks of 16 k
    const int u = wave >> 2, kq = wave & 3, q4 = lane & 3, rsub = lane >> 2;
    const unsigned row_q = (unsigned)t * 32u + 16u * (unsigned)u + (unsigned)rsub;
    const unsigned rowc_q = row_q < rows ? row_q : rows - 1u;
    const unsigned first = (unsigned)t * 32u + 16u * (unsigned)u;
    const unsigned last = first + 15u < rows ? first + 15u : rows - 1u;
    const unsigned s_lo = (first < rows ? first : rows - 1u) / mu, s_hi = last / mu;
    float2 pw[2][4];  // the first 256 {min, max} pairs of slices s_lo and s_lo + 1 (clamped: a repeated pair changes nothing)
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const unsigned sl = s_lo + (unsigned)a < s_hi ? s_lo + (unsigned)a : s_hi;
        const float2* pp = reinterpret_cast<const float2*>(g.partial) + (size_t)sl * g.nblk;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int idx = lane + 64 * j;
            pw[a][j] = pp[idx < g.nblk ? idx : g.nblk - 1];
        }
    }
    float4 xv[8];
    {
        const float* src = g.x + (size_t)rowc_q * 512u + 128u * (unsigned)kq + 4u * (unsigned)q4;
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[c] = *reinterpret_cast<const float4*>(src + 16 * c);
    }
    // FS: the window of v, rows [32 t - pl, 32 t + 31 + FS - 1 - pl] clamped into the tensor (rows of another utterance are fetched and
    // never used), two 1 KiB direct-to-LDS loads a row, spread over the eight waves.  Issued right behind the rows: loads return in
    // order, so the wait for the rows in front of the quantiser below covers them and they cost no register.
    float fw[FS > 0 ? FS : 1], fbv = 0.0f;
    if constexpr (FS > 0) {
        const unsigned tile_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)s_ln;
#pragma unroll
        for (int k2 = 0; k2 < (2 * (32 + FS - 1) + 7) / 8; ++k2) {
            const int it = wave + 8 * k2;  // wave-uniform
            if (it < 2 * (32 + FS - 1)) {
                const int wr = it >> 1, half = it & 1;
                int gr = (int)t * 32 - FS / 2 + wr;
                gr = gr < 0 ? 0 : (gr < (int)rows ? gr : (int)rows - 1);
                rs_dma16(g.fs_x + (size_t)gr * (unsigned)g.fs_pitch + 256 * half + 4 * lane,
                         __builtin_amdgcn_readfirstlane(tile_base + (unsigned)(wr * AS_LN_PITCH * 4 + 1024 * half)));
            }
        }
#pragma unroll
        for (int j = 0; j < FS; ++j) fw[j] = g.fs_w[threadIdx.x * FS + j];
        fbv = (g.fs_b ? g.fs_b : g.fs_w)[threadIdx.x];
    }
    // the column terms of the whole tile row: one column a thread (no branch around a load: a missing bias reads the scale and drops it)
    const int ccol = (int)threadIdx.x < n ? (int)threadIdx.x : n - 1;
    const int csum = epi.col_sums[ccol];
    const float wsv = epi.wscale[epi.wscale_len <= 1 ? 0 : ccol];
    const float bv = (epi.bias ? epi.bias : epi.wscale)[epi.bias ? ccol : 0];
    float lgv = 0.0f, lbv = 0.0f;
    if constexpr (LN) lgv = g.ln_g[threadIdx.x], lbv = g.ln_b[threadIdx.x];  // n == 512 == the workgroup's threads
    __builtin_amdgcn_sched_barrier(0);  // ... and the compiler keeps that order
    // ---- 2. the weights of column tiles 2 wave and 2 wave + 1 (a tile beyond the last one repeats it: multiplied, never stored)
    const int ct0 = TWO ? 2 * wave : (int)blockIdx.y * g.tpw + wave, ct1 = 2 * wave + 1;
    const bool live0 = (TWO || wave < g.tpw) && ct0 < g.nct;  // (a wave without a tile still loads -- clamped -- and quantises; it leaves after the barrier)
    v4i bf0[KS], bf1[TWO ? KS : 1];
    {
        const v4i* w0 = reinterpret_cast<const v4i*>(g.wf) + (size_t)(ct0 < g.nct ? ct0 : g.nct - 1) * KS * 64 + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) bf0[s] = w0[s * 64];
        if constexpr (TWO) {
            const v4i* w1 = reinterpret_cast<const v4i*>(g.wf) + (size_t)(ct1 < g.nct ? ct1 : g.nct - 1) * KS * 64 + lane;
#pragma unroll
            for (int s = 0; s < KS; ++s) bf1[s] = w1[s * 64];
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    s_colsum[threadIdx.x] = csum;
    s_ws[threadIdx.x] = wsv;
    s_bias[threadIdx.x] = epi.bias ? bv : -0.0f;  // x + (-0.0) == x for every x, the sign of zero included
    if constexpr (LN) s_lng[threadIdx.x] = lgv, s_lnb[threadIdx.x] = lbv;
    // ---- 3. the parameters of the slices this wave's rows belong to (wave-uniform loop: one or two in practice), as qrows_kernel<0>
    QParams q = {1.0f, 0.0f, 1.0f, 0};
    {
        const unsigned mine = rowc_q / mu;
        for (unsigned sl = s_lo; sl <= s_hi; ++sl) {
            float mn = 3.40282347e+38f, mx = -3.40282347e+38f;
            const float2* pp = reinterpret_cast<const float2*>(g.partial) + (size_t)sl * g.nblk;
            int i0 = 0;
            if (sl - s_lo < 2u) {  // requested before anything else
                const int a = (int)(sl - s_lo);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float2 w = a == 0 ? pw[0][j] : pw[1][j];
                    mn = w.x < mn ? w.x : mn;
                    mx = w.y > mx ? w.y : mx;
                }
                i0 = 256;
            }
            for (; i0 < g.nblk; i0 += 256) {  // more than 256 pairs a slice, a third slice in 16 rows: fetched now
                float2 w[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int idx = i0 + lane + 64 * j;
                    w[j] = pp[idx < g.nblk ? idx : g.nblk - 1];
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    mn = w[j].x < mn ? w[j].x : mn;
                    mx = w[j].y > mx ? w[j].y : mx;
                }
            }
            mn = wave_allreduce64(mn, [](float cur, float a) { return a < cur ? a : cur; });
            mx = wave_allreduce64(mx, [](float cur, float a) { return a > cur ? a : cur; });
            const QParams qs = make_qparams(mn, mx);
            if (mine == sl) q = qs;
            if (kq == 0 && q4 == 0 && row_q < rows && row_q == sl * mu && blockIdx.y == 0) g.prm[sl] = qs;  // the slice's first row publishes
        }
    }
    // ---- 4. quantise into fragment order: rint(fma(x, 1 / scale, zp)) saturated to u8, minus 128 (K = 512: every element is in the
    //         reference's SIMD body)
    {
        unsigned us = 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            unsigned p = 0u;
            p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(xv[c].x, q.inv_scale, q.zp), 0, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(xv[c].y, q.inv_scale, q.zp), 1, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(xv[c].z, q.inv_scale, q.zp), 2, p);
            p = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf(xv[c].w, q.inv_scale, q.zp), 3, p);
            us = __builtin_amdgcn_sad_u8(p, 0u, us);
            // k-step 4 kq + c / 2; inside its 1 KiB block lane (row, c & 1) holds k = 16 (c & 1) + [0, 16)
            *reinterpret_cast<unsigned*>(s_tile + (4 * kq + (c >> 1)) * 1024 + ((16 * u + rsub) + 32 * (c & 1)) * 16 + 4 * q4) = p ^ 0x80808080u;
        }
        us += (unsigned)__shfl_xor((int)us, 1);  // the row's four lanes: its 128 codes of this quarter of k
        us += (unsigned)__shfl_xor((int)us, 2);
        if (q4 == 0) {
            s_rowsum[kq][16 * u + rsub] = (int)us - 128 * 128;
            if (kq == 0) {
                s_scale[16 * u + rsub] = q.scale;
                s_zp[16 * u + rsub] = q.zp_i;
            }
        }
    }
    __syncthreads();
    if constexpr (FS > 0) {
        // ---- 4b. the memory block, column threadIdx.x, rows top to bottom, in place (dwconv1d_tlc_kernel's arithmetic: taps in
        //          ascending order, taps outside the utterance skipped, FMA chain, bias afterwards, then + v[t])
        const int ch = (int)threadIdx.x;
        constexpr int PL = FS / 2;  // symmetric padding (the host checks): v[t] is window entry PL
        float win[FS];
#pragma unroll
        for (int j = 0; j < FS - 1; ++j) win[j] = s_ln[j * AS_LN_PITCH + ch];
        const int T = (int)mu;
        int tt = (int)(((unsigned)t * 32u) % mu);  // time step of tile row 0 inside its utterance (uniform)
#pragma unroll
        for (int r = 0; r < 32; ++r) {
            win[FS - 1] = s_ln[(r + FS - 1) * AS_LN_PITCH + ch];
            float a = 0.0f;
            if (tt >= PL && tt + (FS - 1 - PL) < T) {  // (uniform) every tap inside the utterance: 161 of 171 rows
#pragma unroll
                for (int j = 0; j < FS; ++j) a = fmaf_(win[j], fw[j], a);
            } else {
#pragma unroll
                for (int j = 0; j < FS; ++j) {
                    const int tq = tt - PL + j;
                    const float f = fmaf_(win[j], fw[j], a);
                    a = (tq >= 0 && tq < T) ? f : a;
                }
            }
            if (g.fs_b) a = a + fbv;
            a = a + win[PL];
            s_ln[r * AS_LN_PITCH + ch] = a;
#pragma unroll
            for (int j = 0; j < FS - 1; ++j) win[j] = win[j + 1];
            tt = tt + 1 == T ? 0 : tt + 1;
        }
        __syncthreads();
    }
    // ---- 5. products: the tile's 16 fragments against both column tiles
    // (the accumulators start from the row / column terms, as in igemm_rs_kernel)
    if (!TWO && !live0) return;  // after the only barrier
    const unsigned row = (unsigned)t * 32u + (unsigned)l31;
    const bool rok = row < rows;
    const unsigned rowc = rok ? row : rows - 1u;
    const float ds = s_scale[l31];
    v16i acc0, acc1;
    {
        const int rowsum = s_rowsum[0][l31] + s_rowsum[1][l31] + s_rowsum[2][l31] + s_rowsum[3][l31];
        const int ca = 128 - s_zp[l31], cbz = 128 - epi.zp_b;
        const int rterm = cbz * rowsum + epi.k * ca * cbz;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const v4i c0 = *reinterpret_cast<const v4i*>(&s_colsum[(ct0 * 32 + 4 * hv + 8 * gq) & 511]);
#pragma unroll
            for (int e = 0; e < 4; ++e) acc0[4 * gq + e] = __mul24(ca, c0[e]) + rterm;
            if constexpr (TWO) {
                const v4i c1 = *reinterpret_cast<const v4i*>(&s_colsum[(ct1 * 32 + 4 * hv + 8 * gq) & 511]);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc1[4 * gq + e] = __mul24(ca, c1[e]) + rterm;
            }
        }
    }
    // the residual operands of both tiles: requested before the products (the rows' registers are free again), used behind them
    float4 ra1[NRES > 0 ? 4 : 1], ra2[NRES > 1 ? 4 : 1], rb1[NRES > 0 ? 4 : 1], rb2[NRES > 1 ? 4 : 1];
    if (NRES > 0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int c0 = ct0 * 32 + 4 * hv + 8 * gq, c1 = ct1 * 32 + 4 * hv + 8 * gq;
            const unsigned at0 = rowc * (unsigned)n + (unsigned)(c0 < n ? c0 : n - 4), at1 = rowc * (unsigned)n + (unsigned)(c1 < n ? c1 : n - 4);
            if constexpr (FS == 0) {
                ra1[gq] = *reinterpret_cast<const float4*>(epi.res1 + at0);
                if (TWO) rb1[gq] = *reinterpret_cast<const float4*>(epi.res1 + at1);
            }
            if (NRES > 1) {
                ra2[gq] = *reinterpret_cast<const float4*>(epi.res2 + at0);
                if (TWO) rb2[gq] = *reinterpret_cast<const float4*>(epi.res2 + at1);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    {
        const v4i* ap = reinterpret_cast<const v4i*>(s_tile) + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i a = ap[s * 64];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf0[s], a, acc0, 0, 0, 0);
            if constexpr (TWO) acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(bf1[s], a, acc1, 0, 0, 0);
        }
    }
    // ---- 6. epilogue (IgemmEpi::value24 with the row terms from LDS), a lane = one row x {4 x 4 columns} of each tile
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)epi.out, 0, (int)(rows * (unsigned)n * 4u), 0x00020000);
    auto finish = [&](const v16i& acc, int ct, const float4 (&res1)[NRES > 0 ? 4 : 1], const float4 (&res2)[NRES > 1 ? 4 : 1]) {
        const unsigned obase = rowc * (unsigned)n + (unsigned)(ct * 32 + 4 * hv);  // rows * n < 2^30
        float smn = 3.40282347e+38f, smx = -3.40282347e+38f;
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int cl = (ct * 32 + 4 * hv + 8 * gq) & 511;
            const float4 ws = *reinterpret_cast<const float4*>(&s_ws[cl]);
            const float4 bs = *reinterpret_cast<const float4*>(&s_bias[cl]);
            auto val = [&](int total, float w, float b) {
                float vf = (float)total;  // _mm256_cvtepi32_ps
                vf = vf * (ds * w);
                vf = vf + b;
                if (RELU) vf = relu0(vf);
                return vf;
            };
            float4 o;
            o.x = val(acc[4 * gq + 0], ws.x, bs.x);
            o.y = val(acc[4 * gq + 1], ws.y, bs.y);
            o.z = val(acc[4 * gq + 2], ws.z, bs.z);
            o.w = val(acc[4 * gq + 3], ws.w, bs.w);
            if (NRES > 0) {
                float4 r1;
                if constexpr (FS > 0) r1 = *reinterpret_cast<const float4*>(&s_ln[l31 * AS_LN_PITCH + ct * 32 + 4 * hv + 8 * gq]);  // the memory block, computed above
                else r1 = res1[gq];
                o.x = o.x + r1.x, o.y = o.y + r1.y, o.z = o.z + r1.z, o.w = o.w + r1.w;
                if (NRES > 1) o.x = o.x + res2[gq].x, o.y = o.y + res2[gq].y, o.z = o.z + res2[gq].z, o.w = o.w + res2[gq].w;
            }
            const bool cok = ct * 32 + 4 * hv + 8 * gq < n;  // n % 4 == 0: a group of four columns is whole or absent
            const v4u bits = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, rok && cok ? (obase + 8u * gq) * 4u : 0xffffffffu, 0, 0);
            if constexpr (LN) *reinterpret_cast<float4*>(&s_ln[l31 * AS_LN_PITCH + ct * 32 + 4 * hv + 8 * gq]) = o;
            if (rok && cok) {
                smn = fminf(smn, fminf(fminf(o.x, o.y), fminf(o.z, o.w)));
                smx = fmaxf(smx, fmaxf(fmaxf(o.x, o.y), fmaxf(o.z, o.w)));
            }
        }
        if (epi.blockstat && ct < g.nct) {  // one {min, max} pair per (row tile, column tile): LeleBuf::rowstat kind 1
            smn = wave_allreduce64(smn, [](float cur, float x) { return x < cur ? x : cur; });
            smx = wave_allreduce64(smx, [](float cur, float x) { return x > cur ? x : cur; });
            if (lane == 0) {
                float* so = epi.blockstat + ((size_t)t * g.nct + ct) * 2;
                so[0] = smn;
                so[1] = smx;
            }
        }
    };
    finish(acc0, ct0, ra1, ra2);
    if constexpr (TWO) finish(acc1, ct1, rb1, rb2);
    if constexpr (LN) {
        __syncthreads();
        as_ln_rows(s_ln, s_lng, s_lnb, (unsigned)t, wave, hv, l31, rows, g.ln_eps, g.ln_out, g.ln_rowstat);
    }
}

// ------------------------------------------------------------------------------------------ K = 2048, N = 512: activations stationary, weights streamed
// The second product of the feed-forward block (5472 x 2048 x 512 on a configs[3] shard) followed by the next layer's LayerNorm.
// igemm_rs_ks4_kernel gives a workgroup one column tile, so no workgroup ever holds a whole row and the LayerNorm was a launch of
// its own (7 us + the sum's round trip).  Here a workgroup owns ONE 32-row tile and ALL 512 columns: the tile's 64 fragment blocks
// (already i8 in fragment order: the quantise pass of the first product wrote them) arrive in LDS by direct-to-LDS loads, every
// wave streams the weight fragments of its two column tiles through an eight-step register ring (128 KiB a wave, the same 1 MiB
// for every workgroup: L2 hits), 2 x 64 products, row sums by v_dot4 on the fragments it reads anyway, then the epilogue of
// igemm_as_kernel -- convert, scale, bias, residuals, 16-byte stores -- and, LN, the LayerNorm phase on the tile in LDS.
struct AskArgs {
    const int8_t* af;      // fragment-major [nrt][64][1024]
    const int8_t* wf;      // fragment-major [16][64][1024]
    unsigned rows;
    const float* ln_g = nullptr;
    const float* ln_b = nullptr;
    float ln_eps = 0.0f;
    float* ln_out = nullptr;
    float* ln_rowstat = nullptr;
};
constexpr int ASK_TILE = 64 * 1024;
constexpr int ASK_LDS = ASK_TILE + AS_LN_LDS;
template <int NRES, bool LN>
__global__ __launch_bounds__(512) void igemm_ask_kernel(AskArgs g, IgemmEpi epi) {
    constexpr int KS = 64, RING = 8;
    extern __shared__ __attribute__((aligned(16))) char ask_lds[];  // [the tile's fragments: 64 KiB][LN: 32 x AS_LN_PITCH floats]
    float* s_ln = reinterpret_cast<float*>(ask_lds + ASK_TILE);
    __shared__ __attribute__((aligned(16))) int s_colsum[512];
    __shared__ __attribute__((aligned(16))) float s_ws[512];
    __shared__ __attribute__((aligned(16))) float s_bias[512];
    __shared__ float s_lng[LN ? 512 : 1], s_lnb[LN ? 512 : 1];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int hv = lane >> 5, l31 = lane & 31;
    const unsigned t = blockIdx.x, rows = g.rows, mu = (unsigned)epi.m;
    constexpr int n = 512;
    // ---- 1. the tile: block b of 64 -> wave b % 8 (eight 1 KiB direct-to-LDS loads a wave), issued before every other load: loads
    //         return in order, so the wait for the column terms below covers them
    {
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)ask_lds;
        const char* src = reinterpret_cast<const char*>(g.af) + ((size_t)t * KS) * 1024 + lane * 16;
#pragma unroll
        for (int j = 0; j < KS / 8; ++j) {
            const int blk = wave + 8 * j;
            rs_dma16(src + (size_t)blk * 1024, __builtin_amdgcn_readfirstlane(base + (unsigned)blk * 1024u));
        }
    }
    // ---- 2. column terms (one column a thread), LayerNorm scale / bias, the row's slice parameters
    const int csum = epi.col_sums[threadIdx.x];
    const float wsv = epi.wscale[epi.wscale_len <= 1 ? 0 : threadIdx.x];
    const float bv = (epi.bias ? epi.bias : epi.wscale)[epi.bias ? threadIdx.x : 0];
    float lgv = 0.0f, lbv = 0.0f;
    if constexpr (LN) lgv = g.ln_g[threadIdx.x], lbv = g.ln_b[threadIdx.x];
    const unsigned row = t * 32u + (unsigned)l31;
    const bool rok = row < rows;
    const unsigned rowc = rok ? row : rows - 1u;
    const unsigned sl = rows == mu ? 0u : rowc / mu;
    const float ds = epi.prm[sl].scale;
    const int zp_i = epi.prm[sl].zp_i;
    __builtin_amdgcn_sched_barrier(0);
    // ---- 3. the first RING steps of this wave's weights (column tiles 2 wave and 2 wave + 1)
    const int ct0 = 2 * wave, ct1 = 2 * wave + 1;
    const v4i* w0 = reinterpret_cast<const v4i*>(g.wf) + (size_t)ct0 * KS * 64 + lane;
    const v4i* w1 = reinterpret_cast<const v4i*>(g.wf) + (size_t)ct1 * KS * 64 + lane;
    // A layer's weights are COLD (70 layers x 1 MiB pass through the caches between two uses) and every workgroup streams the same
    // megabyte: in lock step each of them would meet memory latency at every refill of the ring.  The k-steps of an exact integer sum
    // may be taken in any order, so the workgroups of an XCD (block b runs on XCD b % 8) start at eight different offsets: after the
    // first RING steps every line a workgroup asks for has been fetched by its neighbour already.
    const int rot = (int)((blockIdx.x >> 3) & 7u) * 8;
    v4i ra[RING], rb[RING];
#pragma unroll
    for (int s = 0; s < RING; ++s) ra[s] = w0[((s + rot) & (KS - 1)) * 64], rb[s] = w1[((s + rot) & (KS - 1)) * 64];
    __builtin_amdgcn_sched_barrier(0);
    s_colsum[threadIdx.x] = csum;
    s_ws[threadIdx.x] = wsv;
    s_bias[threadIdx.x] = epi.bias ? bv : -0.0f;
    if constexpr (LN) s_lng[threadIdx.x] = lgv, s_lnb[threadIdx.x] = lbv;
    // the residual operands: requested before the products, used behind them
    float4 ra1[NRES > 0 ? 4 : 1], ra2[NRES > 1 ? 4 : 1], rb1[NRES > 0 ? 4 : 1], rb2[NRES > 1 ? 4 : 1];
    if (NRES > 0) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const unsigned at0 = rowc * (unsigned)n + (unsigned)(ct0 * 32 + 4 * hv + 8 * gq), at1 = rowc * (unsigned)n + (unsigned)(ct1 * 32 + 4 * hv + 8 * gq);
            ra1[gq] = *reinterpret_cast<const float4*>(epi.res1 + at0);
            rb1[gq] = *reinterpret_cast<const float4*>(epi.res1 + at1);
            if (NRES > 1) {
                ra2[gq] = *reinterpret_cast<const float4*>(epi.res2 + at0);
                rb2[gq] = *reinterpret_cast<const float4*>(epi.res2 + at1);
            }
        }
    }
    __syncthreads();  // every wave's part of the tile has landed (its own loads behind it have been waited for above)
    // ---- 4. 2 x 64 products; the weights of step s + RING are requested as step s's registers fall free
    v16i acc0, acc1;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc0[i] = 0, acc1[i] = 0;
    int rsp = 0;
    {
        const v4i* ap = reinterpret_cast<const v4i*>(ask_lds) + lane;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i a = ap[((s + rot) & (KS - 1)) * 64];
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(ra[s % RING], a, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(rb[s % RING], a, acc1, 0, 0, 0);
            rsp = __builtin_amdgcn_sdot4(a[0], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[1], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[2], 0x01010101, rsp, false);
            rsp = __builtin_amdgcn_sdot4(a[3], 0x01010101, rsp, false);
            if (s + RING < KS) ra[s % RING] = w0[((s + RING + rot) & (KS - 1)) * 64], rb[s % RING] = w1[((s + RING + rot) & (KS - 1)) * 64];
        }
    }
    // ---- 5. epilogue: IgemmEpi::value24 as igemm_rs_ks4_kernel applies it (row terms added behind the products)
    const int rowsum = rsp + __shfl_xor(rsp, 32);  // the two k halves of the row
    const int ca = 128 - zp_i, cbz = 128 - epi.zp_b;
    const int rterm = cbz * rowsum + epi.k * ca * cbz;
    const auto out_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)epi.out, 0, (int)(rows * (unsigned)n * 4u), 0x00020000);
    auto finish = [&](const v16i& acc, int ct, const float4 (&res1)[NRES > 0 ? 4 : 1], const float4 (&res2)[NRES > 1 ? 4 : 1]) {
        const unsigned obase = rowc * (unsigned)n + (unsigned)(ct * 32 + 4 * hv);
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int cl = ct * 32 + 4 * hv + 8 * gq;
            const float4 ws = *reinterpret_cast<const float4*>(&s_ws[cl]);
            const float4 bs = *reinterpret_cast<const float4*>(&s_bias[cl]);
            const v4i cs = *reinterpret_cast<const v4i*>(&s_colsum[cl]);
            auto val = [&](int a, int c, float w, float b) {
                float vf = (float)(a + rterm + __mul24(ca, c));  // _mm256_cvtepi32_ps
                vf = vf * (ds * w);
                vf = vf + b;
                if (epi.relu) vf = relu0(vf);
                return vf;
            };
            float4 o;
            o.x = val(acc[4 * gq + 0], cs[0], ws.x, bs.x);
            o.y = val(acc[4 * gq + 1], cs[1], ws.y, bs.y);
            o.z = val(acc[4 * gq + 2], cs[2], ws.z, bs.z);
            o.w = val(acc[4 * gq + 3], cs[3], ws.w, bs.w);
            if (NRES > 0) {
                o.x = o.x + res1[gq].x, o.y = o.y + res1[gq].y, o.z = o.z + res1[gq].z, o.w = o.w + res1[gq].w;
                if (NRES > 1) o.x = o.x + res2[gq].x, o.y = o.y + res2[gq].y, o.z = o.z + res2[gq].z, o.w = o.w + res2[gq].w;
            }
            const v4u bits = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            __builtin_amdgcn_raw_buffer_store_b128(bits, out_rsrc, rok ? (obase + 8u * gq) * 4u : 0xffffffffu, 0, 0);
            if constexpr (LN) *reinterpret_cast<float4*>(&s_ln[l31 * AS_LN_PITCH + cl]) = o;
        }
    };
    finish(acc0, ct0, ra1, ra2);
    finish(acc1, ct1, rb1, rb2);
    if constexpr (LN) {
        __syncthreads();
        as_ln_rows(s_ln, s_lng, s_lnb, t, wave, hv, l31, rows, g.ln_eps, g.ln_out, g.ln_rowstat);
    }
}

}  // namespace
