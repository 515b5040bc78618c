// frontend.hip -- SenseVoiceFrontend on gfx950: PCM -> (x32768, mean, pre-emphasis, Hann) -> 512-pt FFT ->
// power -> sparse mel -> ln -> LFR, one launch pair per batch of utterances.
//
// Replaces /root/reference/src/features/pipeline.rs:38-193 (+ window.rs, fft.rs, mel.rs, lfr.rs) and
// /root/reference/src/kernels/fft.rs:51-266.
//
// Numerics contract: the power spectrum is BIT-EXACT with the reference's x86 path (see fe_core.h for why a
// merely "accurate" FFT is not enough); the mel sum uses the reference's order and roundings; only logf may
// differ in the last ulp.  Two kernels:
//   fe_frame_sum_kernel : the per-frame mean is a sequential f32 sum of 400 samples (pipeline.rs:115-116);
//                         reproduced exactly by giving each LANE one frame (64 independent chains per wave)
//                         and transposing the PCM tile through LDS so every step is a conflict-free row read.
//   fe_main_kernel      : 16 lanes per frame, 4 frames per wave pass; stages 1-5 in registers, one LDS
//                         exchange (two 2 KB rounds), stages 6-9 in registers, power -> LDS (aliasing the
//                         exchange buffer) -> lane-per-filter sparse mel -> ln -> LFR scatter store.
// HBM traffic per utterance: PCM read once by each kernel (the second read is an L2/MALL hit for batches
// below the 256 MiB Infinity Cache) + the LFR output; algorithmic bytes = 4*S + 4*T*560.
#include <type_traits>
#include "common.h"
#include "fe_core.h"

#include <math.h>

#include <algorithm>

using namespace lele;

namespace {

constexpr int kMaxMelRounds = 8;  // n_mels <= 128 on the fast path

struct FeDev {                 // device tables (all owned by LeleFrontend)
    const float* tw_re;        // 511 (+1 pad) reference twiddle table
    const float* tw_im;
    const float2* tw;          // same, interleaved, for LDS staging
    const float* window;       // [frame_len]
    const float* melw;         // [total_steps][16]: weight of step t for lane pm
    const int* mel_start;      // [rounds*16] first bin of filter m (0 for padding filters)
    const int* mel_step_off;   // [rounds+1] first step of each round in melw
    int mel_rounds;            // ceil(n_mels/16)
    int mel_steps;             // total steps (rows of melw)
    int n_mels;
    int lfr_m, lfr_n;
};

// ------------------------------------------------------------------------------------------------------
// Kernel 1: exact sequential frame sums.  grid (ceil(F/64), batch), block 64.
// ------------------------------------------------------------------------------------------------------
constexpr int kSumRows = 66;     // hop-sized rows covered by 64 consecutive frames (64 + 2 of overlap, 400 = 2.5 hops)
constexpr int kSumPitch = 164;   // floats per row (656 B, 16-B aligned): lane l reads row l(+1,+2) with ds_read_b128
                                 // at word 164*l = 36*l (mod 64) -> a 16-lane group covers all 64 banks exactly once

__global__ __launch_bounds__(64) void fe_frame_sum_kernel(const float* __restrict__ pcm, int64_t utt_stride,
                                                          int64_t pcm_len, int64_t num_frames,
                                                          float* __restrict__ mean_out, int aligned16) {
    // The 64 frames of this block span one contiguous run of samples; it is read from HBM exactly once
    // (linear, coalesced float4) and laid out as [hop row][offset in hop]: frame i = rows i, i+1 and half of i+2.
    __shared__ __attribute__((aligned(16))) float tile[kSumRows * kSumPitch];
    const int lane = threadIdx.x;
    const int64_t f0 = (int64_t)blockIdx.x * 64;
    const float* base = pcm + (int64_t)blockIdx.y * utt_stride;
    const int64_t s0 = f0 * fe::kHop;  // first sample of the run
    constexpr int kVecPerRow = fe::kHop / 4;
    constexpr int kVecs = kSumRows * kVecPerRow;
    constexpr int kIters = (kVecs + 63) / 64;  // 42 float4 per lane
    constexpr int kBatch = 14;                  // loads kept in flight together (56 VGPRs)
    static_assert(kIters % kBatch == 0, "batching");
    for (int b0 = 0; b0 < kIters; b0 += kBatch) {
        float4 st[kBatch];
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = (b0 + u) * 64 + lane;
            int64_t s = s0 + 4 * (int64_t)idx;
            // clamp instead of branching so that every load of the batch is issued before the first wait; lanes
            // that would read past the utterance only ever feed frames >= num_frames, whose sums are discarded
            if (s + 3 >= pcm_len) s = aligned16 ? ((pcm_len - 4) & ~int64_t(3)) : (pcm_len - 4);
            if (s < 0) s = 0;
            const float* src = base + s;
            if (aligned16) {
                st[u] = *reinterpret_cast<const float4*>(src);
            } else {
                st[u] = make_float4(src[0], src[1], src[2], src[3]);
            }
        }
#pragma unroll
        for (int u = 0; u < kBatch; ++u) {
            const int idx = (b0 + u) * 64 + lane;
            if (idx < kVecs) {
                float4 xv = st[u];
                // 1. Scale (pipeline.rs:90-112): exact multiplication by 2^15
                xv.x *= 32768.0f;
                xv.y *= 32768.0f;
                xv.z *= 32768.0f;
                xv.w *= 32768.0f;
                const int row = idx / kVecPerRow, c = idx - row * kVecPerRow;
                *reinterpret_cast<float4*>(&tile[row * kSumPitch + 4 * c]) = xv;
            }
        }
    }
    __syncthreads();
    // 2. raw_frame.iter().sum() (pipeline.rs:115): one lane per frame, 400 adds in index order
    float sum = 0.0f;
#pragma unroll
    for (int seg = 0; seg < 3; ++seg) {
        const float* rowp = &tile[(lane + seg) * kSumPitch];
        const int nvec = seg < 2 ? kVecPerRow : (fe::kFrame - 2 * fe::kHop) / 4;
#pragma unroll 10
        for (int k = 0; k < nvec; ++k) {
            const float4 r = *reinterpret_cast<const float4*>(rowp + 4 * k);
            sum = sum + r.x;
            sum = sum + r.y;
            sum = sum + r.z;
            sum = sum + r.w;
        }
    }
    const int64_t f = f0 + lane;
    if (f < num_frames) mean_out[(int64_t)blockIdx.y * num_frames + f] = sum / (float)fe::kFrame;  // pipeline.rs:116
}

// ------------------------------------------------------------------------------------------------------
// Kernel 2: FFT + mel + log + LFR.  grid (ceil(F/64), batch), block 256 = 4 independent waves of 16 frames.
// ------------------------------------------------------------------------------------------------------
constexpr int kFrameXchgFloats = 2 * 16 * fe::kXchgPitch;  // 544 floats = 2176 B: odd frames start 32 banks later
constexpr int kWaveLdsFloats = 4 * kFrameXchgFloats;  // 8704 B per wave
constexpr int kPStride = 272;          // power row stride (words); 4*272 <= kWaveLdsFloats.  = 16 mod 32: the two frames of a half-wave write disjoint banks
                                       // and the lane-per-filter reads collide least (model of the 80-filter bank over all pitches: 186 cycles a pass, 258: 236)

struct TwLds {  // twiddle accessor over the LDS copy (phase B: per-lane indices)
    const float2* t;
    __device__ __forceinline__ float2 at(int i) const { return t[i]; }
};

template <int MODE>
__device__ __forceinline__ float lane_rot_prev(float v, int p) {
    // value of lane (p-1) mod 16 inside this 16-lane row
    if (MODE == 1) {
        // DPP row_ror:1 -- data moves to the next higher lane, lane 0 receives lane 15
        return __builtin_bit_cast(float,
                                  __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, true));
    } else {
        return __shfl(v, (p + 15) & 15, 16);
    }
}

// STDMEL: the mel bank has the round structure of the default config (80 mels @16 kHz/512: 5 rounds of
// 3,4,6,10,17 steps, no filter reaching past bin 256) -- verified on the host -- so the lane-per-filter dot
// products are fully unrolled with immediate LDS offsets; otherwise the table-driven loop is used.
constexpr int kStdMelLen[5] = {3, 4, 6, 10, 17};
constexpr int kStdMelOff[6] = {0, 3, 7, 13, 23, 40};

// FUSED: the block first stages its run of PCM (64 frames = 10 480 samples, read from HBM once) into an LDS tile that
// aliases the exchange region, wave 0 computes the 64 exact sequential frame sums from it (as fe_frame_sum_kernel
// does), and the per-pass sample loads below then hit L2 -- no separate sum kernel, no second HBM read of the PCM.
// PASSES: frames per workgroup = 16 * PASSES (a wave takes 4 frames a pass).  LOWREG: the 25 window coefficients of a lane are
// re-read from the (L1-resident) table every pass instead of living in registers.  Four waves per SIMD (<= 128 registers, PASSES = 3 so
// that four workgroups' LDS fits a CU) were measured twice in round 4: <3, true> with 48 spilled registers 7.41 ms per 2048 x 30 s, and
// -- after the pass body lost its copies -- <3, false> at 128 registers without a spill in the loop 3.59 ms against 3.47 ms for
// <4, false> at three waves (148 registers) in the same call: the product instantiates <4, false> only; docs/experiments.md.
template <int MODE, bool STDMEL, bool FUSED, int PASSES, bool LOWREG>
__device__ __forceinline__ void fe_main_body(const float* __restrict__ pcm, int64_t utt_stride,
                                             int64_t num_frames, int64_t t_lfr,
                                             const float* __restrict__ means, const FeDev& tb,
                                             float* __restrict__ out, float* __restrict__ logmel_out,
                                             int aligned16, float* s_melw) {
    constexpr int FR = 16 * PASSES;           // frames per workgroup
    constexpr int kRows = FR + 2;             // hop rows of the run (400 = 2.5 hops)
    constexpr int kXFloats = FUSED ? (kRows * kSumPitch > 4 * kWaveLdsFloats ? kRows * kSumPitch : 4 * kWaveLdsFloats)
                                   : 4 * kWaveLdsFloats;
    __shared__ float2 s_tw[LOWREG ? 480 : 512];   // LOWREG: only the entries phase B reads (stages 6..9: 31 .. 510)
    __shared__ __attribute__((aligned(16))) float s_x[kXFloats];
    __shared__ float s_mean[64];
    __shared__ int s_mstart[16 * (STDMEL ? 5 : kMaxMelRounds)];
    __shared__ int s_moff[kMaxMelRounds + 1];
    constexpr int kTwBase = LOWREG ? 31 : 0;

    for (int i = threadIdx.x; i < 511 - kTwBase; i += 256) s_tw[i] = tb.tw[i + kTwBase];
    if (!STDMEL)
        for (int i = threadIdx.x; i < tb.mel_steps * 16; i += 256) s_melw[i] = tb.melw[i];
    if (threadIdx.x < tb.mel_rounds * 16) s_mstart[threadIdx.x] = tb.mel_start[threadIdx.x];
    if (threadIdx.x <= tb.mel_rounds) s_moff[threadIdx.x] = tb.mel_step_off[threadIdx.x];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, p = lane & 15;
    if (FUSED) {
        // stage the run [64*blockIdx.x*hop, +66 hops) as [hop row][offset] (x32768, exact) -- same layout and bank
        // argument as fe_frame_sum_kernel, 256 threads, all loads of a thread issued before the first use
        const float* ubase = pcm + (int64_t)blockIdx.y * utt_stride;
        const int64_t s0 = (int64_t)blockIdx.x * FR * fe::kHop;
        constexpr int kVecPerRow = fe::kHop / 4;
        constexpr int kVecs = kRows * kVecPerRow;
        constexpr int kIters = (kVecs + 255) / 256;
        float4 st[kIters];
#pragma unroll
        for (int u = 0; u < kIters; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            int64_t sidx = s0 + 4 * (int64_t)idx;
            if (sidx + 3 >= utt_stride) sidx = aligned16 ? ((utt_stride - 4) & ~int64_t(3)) : (utt_stride - 4);
            if (sidx < 0) sidx = 0;
            const float* src = ubase + sidx;
            if (aligned16)
                st[u] = *reinterpret_cast<const float4*>(src);
            else
                st[u] = make_float4(src[0], src[1], src[2], src[3]);
        }
#pragma unroll
        for (int u = 0; u < kIters; ++u) {
            const int idx = u * 256 + (int)threadIdx.x;
            if (idx < kVecs) {
                float4 xv = st[u];
                xv.x *= 32768.0f;
                xv.y *= 32768.0f;
                xv.z *= 32768.0f;
                xv.w *= 32768.0f;
                const int row = idx / kVecPerRow, c = idx - row * kVecPerRow;
                *reinterpret_cast<float4*>(&s_x[row * kSumPitch + 4 * c]) = xv;
            }
        }
        __syncthreads();
    }
    // window coefficients and the first pass's samples are requested before the sums / barrier below, so that their
    // latency overlaps the 400-step add chain of wave 0
    const int h = fe::rev4(p);
    float* xw = s_x + wave * kWaveLdsFloats;           // this wave's exchange / power region
    float* xf = xw + g * kFrameXchgFloats;             // this frame's 2 KB round buffer
    const float* base = pcm + (int64_t)blockIdx.y * utt_stride;
    const float* mean_u = means + (int64_t)blockIdx.y * num_frames;
    const int d_out = tb.n_mels * tb.lfr_m;
    float* out_u = out ? out + (int64_t)blockIdx.y * t_lfr * d_out : nullptr;
    float* lm_u = logmel_out ? logmel_out + (int64_t)blockIdx.y * num_frames * tb.n_mels : nullptr;

    // Hann window coefficients of this lane's samples n = p + 16 q (features/window.rs), kept in registers (LOWREG: re-read per pass)
    float win[fe::kQ];
    if (!LOWREG) {
#pragma unroll
        for (int q = 0; q < fe::kQ; ++q) win[q] = tb.window[p + 16 * q];
    }

    // raw PCM of the next pass is fetched while the current pass computes (software prefetch): the loads are
    // unconditional (clamped to frame 0 for the tail) so that all 25 are in flight together.
    float xr[fe::kQ];
    float mean_next = 0.0f;
    auto issue_loads = [&](int pass) {
        const int64_t f = (int64_t)blockIdx.x * FR + wave * (4 * PASSES) + pass * 4 + g;
        const int64_t fc = f < num_frames ? f : 0;
        const float* src = base + fc * fe::kHop + p;
        if (!FUSED) mean_next = mean_u[fc];
#pragma unroll
        for (int q = 0; q < fe::kQ; ++q) xr[q] = src[16 * q];
    };
    issue_loads(0);
    if (FUSED) {
        constexpr int kVecPerRow = fe::kHop / 4;
        if (wave == 0 && lane < FR) {  // raw_frame.iter().sum() (pipeline.rs:115): one lane per frame, 400 adds in index order
            float sum = 0.0f;
#pragma unroll
            for (int seg = 0; seg < 3; ++seg) {
                const float* rowp = &s_x[(lane + seg) * kSumPitch];
                const int nvec = seg < 2 ? kVecPerRow : (fe::kFrame - 2 * fe::kHop) / 4;
#pragma unroll 10
                for (int k = 0; k < nvec; ++k) {
                    const float4 r = *reinterpret_cast<const float4*>(rowp + 4 * k);
                    sum = sum + r.x;
                    sum = sum + r.y;
                    sum = sum + r.z;
                    sum = sum + r.w;
                }
            }
            s_mean[lane] = sum / (float)fe::kFrame;  // pipeline.rs:116
        }
    }
    __syncthreads();  // tables (and, when FUSED, the means) are in LDS; the PCM tile may now be overwritten

    // One pass = 4 frames per wave.  The last pass is a second copy of the body WITHOUT the request for the next pass's samples: with
    // that request behind a branch the compiler sank half of the pre-emphasis below it -- the lane rotation and its multiply ended up
    // in different blocks (no DPP operand: 25 v_mov_dpp + 25 v_mov a pass) and the 25 samples were copied twice per pass.
    auto run_pass = [&](int pass, auto load_next_c) {
        constexpr bool load_next = decltype(load_next_c)::value;
        const int64_t f = (int64_t)blockIdx.x * FR + wave * (4 * PASSES) + pass * 4 + g;
        const bool valid = f < num_frames;
        const float mean = FUSED ? s_mean[wave * (4 * PASSES) + pass * 4 + g] : mean_next;
        if (LOWREG) {  // the compiler must not hoist these loads out of the pass loop (that is the register form again)
            int off = 0;
            asm volatile("" : "+v"(off));
#pragma unroll
            for (int q = 0; q < fe::kQ; ++q) win[q] = tb.window[p + 16 * q + off];
        }

        // 1./2. scale and mean subtraction (pipeline.rs:90-137): x * 32768 is exact (a power of two), so the product followed by the
        // subtraction rounds once -- which is what ONE fused multiply-add does: the same bits in half the instructions
        float v[fe::kQ];
        const float neg_mean = -mean;
#pragma unroll
        for (int q = 0; q < fe::kQ; ++q) v[q] = __builtin_fmaf(xr[q], 32768.0f, neg_mean);
        // next pass's samples: requested as early as the registers are free -- the workgroup's tile is still in its XCD's L2 then (requested
        // half a pass later, after phase A: the same time, 9 % more fabric traffic)
        if (load_next) issue_loads(pass + 1);
        // 3. pre-emphasis (pipeline.rs:140-142): y[n] = v[n] - 0.97*v[n-1] for n >= 1; v[n-1] lives in lane p-1
        //    (same q) or, for p == 0, in lane 15 at q-1.  4. window (pipeline.rs:145-166).
        float xin[fe::kRegs];
#pragma unroll
        for (int r = 0; r < fe::kRegs; ++r) xin[r] = 0.0f;
        //    The select happens BEFORE the rotation, on lane 15 (it sends v[q - 1], every other lane v[q]): the rotation then has one
        //    consumer and becomes that multiply's DPP operand (v_mul_f32_dpp; 0.97 in a register, a DPP instruction takes no literal)
        float k97;
        asm("v_mov_b32 %0, 0x3f7851ec" : "=v"(k97));  // 0.97f
#pragma unroll
        for (int q = 0; q < fe::kQ; ++q) {
            const float send = (p == 15 && q > 0) ? v[q > 0 ? q - 1 : 0] : v[q];
            const float prev = lane_rot_prev<MODE>(send, p);
            float y = fe::fsub(v[q], fe::fmul(k97, prev));
            if (q == 0) y = (p == 0) ? v[0] : y;  // sample 0 is not pre-emphasised
            xin[fe::rev5(q)] = fe::fmul(y, win[q]);
        }

        // 5. FFT stages 1..5 (wave-uniform twiddles straight from the table)
        fe::cf a[fe::kRegs];
        fe::phase_a_fast(xin, a, tb.tw_re, tb.tw_im);

        float pw[2][8];
        float p256 = 0.0f;
#pragma unroll
        for (int rho = 0; rho < 2; ++rho) {
            // exchange: register r = rho*16 + cc of lane h  ->  lane cc, register v = h
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int cc = 0; cc < 16; ++cc) {
                const int slot = fe::xchg_slot(h, cc);
                *reinterpret_cast<float2*>(xf + 2 * slot) = make_float2(a[rho * 16 + cc].x, a[rho * 16 + cc].y);
            }
            __builtin_amdgcn_wave_barrier();
            fe::cf bq[16];
#pragma unroll
            for (int vv = 0; vv < 16; ++vv) {
                const float2 t = *reinterpret_cast<const float2*>(xf + 2 * fe::xchg_slot(vv, p));
                bq[vv] = fe::cmk(t.x, t.y);
            }
            __builtin_amdgcn_wave_barrier();
            // stages 6..9 with per-lane twiddles from LDS
            float re256 = 0.0f;
            fe::phase_b(bq, p, rho, [&](int i) { const float2 w = s_tw[i - kTwBase]; return fe::cmk(w.x, w.y); }, &re256);
            if (rho == 0) p256 = re256;  // position 256 (used by lane p == 0)
            // 6. power spectrum (pipeline.rs:165-169); bins 0 and 256 have im forced to 0 (kernels/fft.rs:256-261)
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) {
                fe::cf z = bq[vv];
                if (vv == 0 && rho == 0) z = fe::cmk(z.x, (p == 0) ? 0.0f : z.y);
                pw[rho][vv] = fe::power(z);
            }
        }
        // power -> LDS, row layout [frame g][bin]; aliases the exchange region (all exchange reads are done)
        float* prow = xw + g * kPStride;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int rho = 0; rho < 2; ++rho)
#pragma unroll
            for (int vv = 0; vv < 8; ++vv) prow[(2 * vv + rho) * 16 + p] = pw[rho][vv];
        if (p == 0) prow[256] = fe::power(p256, 0.0f);
        __builtin_amdgcn_wave_barrier();

        // LFR targets of this frame (lfr.rs:36-52): rows i, blocks b with n*i + b - pad == f.  Computed once per
        // frame in 32-bit arithmetic; the clamped first/last frames are handled in the rare branches below.
        const int fi = (int)f, nfr = (int)num_frames, tl = (int)t_lfr;
        const int pad = (tb.lfr_m - 1) / 2;
        const int fp = fi + pad;
        const int b_first = fp % tb.lfr_n, i_first = fp / tb.lfr_n;

        // 7. sparse mel: lane pm = p owns filters m = 16*rd + p; sequential sum += w*P (mel.rs:92-104)
        // 8. ln(max(x, 1e-5)) (mel.rs:124-128) and LFR scatter (lfr.rs:18-54)
        float vals[STDMEL ? 5 : kMaxMelRounds];
#pragma unroll
        for (int rd = 0; rd < (STDMEL ? 5 : kMaxMelRounds); ++rd) {
            vals[rd] = 0.0f;
            if (!STDMEL && rd >= tb.mel_rounds) continue;
            const int m = rd * 16 + p;
            const int start = s_mstart[m];
            const float* pp = prow + start;
            float acc = 0.0f;
            if (STDMEL) {
                const float* wp = tb.melw + kStdMelOff[rd < 5 ? rd : 0] * 16 + p;  // L1-resident table, coalesced
                float w[17];
#pragma unroll
                for (int t = 0; t < 17; ++t)
                    if (t < kStdMelLen[rd < 5 ? rd : 0]) w[t] = wp[t * 16];
#pragma unroll
                for (int t = 0; t < 17; ++t)
                    if (t < kStdMelLen[rd < 5 ? rd : 0]) acc = fe::fadd(acc, fe::fmul(w[t], pp[t]));
            } else {
                const int t0 = s_moff[rd], t1 = s_moff[rd + 1];
                const float* wp = s_melw + t0 * 16 + p;
                const int lim = 256 - start;
#pragma unroll 4
                for (int t = 0; t < t1 - t0; ++t) {
                    const float pv = pp[t < lim ? t : lim];
                    acc = fe::fadd(acc, fe::fmul(wp[t * 16], pv));
                }
            }
            // ln(x) = log2(x) * ln2 with ln2 as hi + lo (what __logf expands to: v_log_f32, then the product in extended precision), without
            // __logf's rescaling of denormal inputs and its infinity test (compare, two selects, ldexp, a subtraction of 0, compare,
            // select): the input is in [1e-5, f32 max) here, so the bits are the same in 5 instructions instead of 12 (a mel sum that
            // overflowed to +inf -- samples beyond 1e9 -- gives NaN here and +inf there)
            {
                const float l2 = __builtin_amdgcn_logf(fmaxf(acc, 1e-5f));
                const float hi = __builtin_bit_cast(float, 0x3f317217u), lo = __builtin_bit_cast(float, 0x3377d1cfu);
                const float r = fe::fmul(l2, hi);
                vals[rd] = fe::fadd(r, fe::ffma(l2, lo, fe::ffma(l2, hi, -r)));
            }
        }
        // store: log-mel row (test hook) and/or the LFR scatter; lane p writes mels p, p+16, ... of each target
        if (valid) {
            constexpr int kR = STDMEL ? 5 : kMaxMelRounds;
            auto put = [&](float* dst) {
#pragma unroll
                for (int rd = 0; rd < kR; ++rd)
                    if (rd * 16 + p < tb.n_mels) __builtin_nontemporal_store(vals[rd], &dst[rd * 16]);  // streamed: keep L2 for the PCM
            };
            if (lm_u) put(lm_u + (int64_t)fi * tb.n_mels + p);
            if (out_u) {
                int i = i_first;
                for (int b = b_first; b < tb.lfr_m && i >= 0; b += tb.lfr_n, --i)
                    if (i < tl) put(out_u + i * d_out + b * tb.n_mels + p);
                if (fi == 0) {  // left clamp: raw index < 0 reads frame 0
                    for (int i2 = 0; i2 * tb.lfr_n - pad < 0 && i2 < tl; ++i2)
                        for (int b = 0; b < tb.lfr_m && i2 * tb.lfr_n + b - pad < 0; ++b)
                            put(out_u + i2 * d_out + b * tb.n_mels + p);
                }
                if (fi == nfr - 1) {  // right clamp: raw index > T-1 reads the last frame
                    for (int i2 = tl - 1; i2 >= 0 && i2 * tb.lfr_n + (tb.lfr_m - 1) - pad > fi; --i2)
                        for (int b = tb.lfr_m - 1; b >= 0 && i2 * tb.lfr_n + b - pad > fi; --b)
                            put(out_u + i2 * d_out + b * tb.n_mels + p);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll 1
    for (int pass = 0; pass + 1 < PASSES; ++pass) run_pass(pass, std::true_type{});
    run_pass(PASSES - 1, std::false_type{});
}

template <int MODE, bool STDMEL, bool FUSED>
__global__ __launch_bounds__(256) void fe_main_kernel(const float* __restrict__ pcm, int64_t utt_stride, int64_t num_frames, int64_t t_lfr,
                                                      const float* __restrict__ means, FeDev tb, float* __restrict__ out,
                                                      float* __restrict__ logmel_out, int aligned16) {
    extern __shared__ float s_melw[];  // [mel_steps][16] (the table-driven mel loop only)
    fe_main_body<MODE, STDMEL, FUSED, 4, false>(pcm, utt_stride, num_frames, t_lfr, means, tb, out, logmel_out, aligned16, s_melw);
}
// ------------------------------------------------------------------------------------------------------
// Generic path: any FeatureConfig the reference accepts (other sample rates / frame lengths / n_fft = 1024).
// The same operations in the same order as pipeline.rs:84-187, spread over four simple kernels + the generic
// radix-2 FFT of features_ops.hip; nothing is fused, it only has to be bit-exact.
// ------------------------------------------------------------------------------------------------------
// (rows = utterances x frames: row r is frame r % num_frames of utterance r / num_frames, utterances pcm_len samples apart)
__global__ void fe_generic_mean_kernel(const float* __restrict__ pcm, int64_t rows, int64_t num_frames, int64_t pcm_len, int frame_len, int hop,
                                       float* __restrict__ mean) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= rows) return;
    const float* src = pcm + (f / num_frames) * pcm_len + (f % num_frames) * hop;
    float sum = 0.0f;
    int j = 0;
    if ((((uintptr_t)src) & 15) == 0) {   // sixteen samples requested (as four 16-byte loads) before their additions, which stay sequential
        for (; j + 16 <= frame_len; j += 16) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + j + 4 * u);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                sum = sum + v[u].x * 32768.0f;
                sum = sum + v[u].y * 32768.0f;
                sum = sum + v[u].z * 32768.0f;
                sum = sum + v[u].w * 32768.0f;
            }
        }
    }
    for (; j < frame_len; ++j) sum = sum + src[j] * 32768.0f;  // raw_frame.iter().sum(), pipeline.rs:115
    mean[f] = sum / (float)frame_len;
}
__global__ void fe_generic_frame_kernel(const float* __restrict__ pcm, const float* __restrict__ mean,
                                        const float* __restrict__ window, int64_t rows, int64_t num_frames, int64_t pcm_len, int frame_len, int hop,
                                        int n_fft, float* __restrict__ frames) {
    const int64_t total = rows * n_fft;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / n_fft;
        const int j = (int)(i - f * n_fft);
        float v = 0.0f;
        if (j < frame_len) {
            const float* src = pcm + (f / num_frames) * pcm_len + (f % num_frames) * hop;
            const float m = mean[f];
            const float cur = fe::fsub(fe::fmul(src[j], 32768.0f), m);
            float y = cur;
            if (j >= 1) y = fe::fsub(cur, fe::fmul(0.97f, fe::fsub(fe::fmul(src[j - 1], 32768.0f), m)));  // pipeline.rs:140-142
            v = fe::fmul(y, window[j]);
        }
        frames[i] = v;
    }
}
__global__ void fe_generic_mel_kernel(const float* __restrict__ power, int64_t num_frames, int n_bins, int n_mels,
                                      const int* __restrict__ mstart, const int* __restrict__ moff,
                                      const float* __restrict__ mw, float* __restrict__ logmel) {
    const int64_t total = num_frames * n_mels;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t f = i / n_mels;
        const int m = (int)(i - f * n_mels);
        const float* p = power + f * n_bins + mstart[m];
        float acc = 0.0f;
        for (int t = moff[m]; t < moff[m + 1]; ++t) acc = fe::fadd(acc, fe::fmul(mw[t], p[t - moff[m]]));  // mel.rs:92-104
        logmel[i] = __logf(fmaxf(acc, 1e-5f));                                                            // mel.rs:124-128
    }
}
// The composed path's first four kernels as ONE (every FeatureConfig other than 400 / 160 / 512 ran frame means |
// pre-emphasis + window | radix-2 FFT | sparse mel + log as four launches over [rows, n_fft] and [rows, bins] scratch tensors --
// 10-20 x the fused default).  A workgroup takes FPB consecutive frames of one utterance: their PCM span goes to LDS once, one lane a
// frame forms the frame sum in the reference's order (pipeline.rs:115: a sequential f32 sum), all threads write the windowed frames
// bit-reversed into LDS, run the butterfly network of fft.rs:172-266 stage by stage there (twiddles in LDS), form the power spectrum
// and the sparse mel sums (mel.rs:92-104) out of LDS and store ln(max(x, 1e-5)).  The SAME operations in the SAME order as the four
// kernels (the fe:: helpers of fe_core.h): bit-identical log-mel rows; nothing but the log-mel leaves the workgroup.
#define FE_PAD(i) ((i) + ((i) >> 5))
__global__ __launch_bounds__(256) void fe_generic_fused_kernel(const float* __restrict__ pcm, int64_t pcm_len, int64_t num_frames, int frame_len,
                                                               int hop, int n, int log2n, int fpb, const float* __restrict__ window,
                                                               const float* __restrict__ tw_re_g, const float* __restrict__ tw_im_g, int n_mels,
                                                               const int* __restrict__ mstart, const int* __restrict__ moff,
                                                               const float* __restrict__ mw, const float* __restrict__ means,
                                                               float* __restrict__ logmel) {
    extern __shared__ float gfl[];
    const int nb = n / 2 + 1, span = (fpb - 1) * hop + frame_len;
    // re / im [fpb][n], one pad word per 32 (FE_PAD): the register passes read 2^R points a thread at strides of 1, 8, 64, ...
    float* re = gfl;
    float* im = re + FE_PAD(fpb * n) + 1;
    float* pw = im + FE_PAD(fpb * n) + 1;  // [fpb][nb]
    float* sp = pw + fpb * nb;             // [span] samples, then [fpb] means
    float* smean = sp + span;
    const int tid = threadIdx.x;
    const int64_t f0 = (int64_t)blockIdx.x * fpb;
    const int nfr = (int)(num_frames - f0 < fpb ? num_frames - f0 : fpb);
    const float* src = pcm + (int64_t)blockIdx.y * pcm_len + f0 * hop;
    const int have = (nfr - 1) * hop + frame_len;   // samples of the frames that exist (every frame lies inside the utterance)
    for (int i = tid; i < span; i += 256) sp[i] = i < have ? src[i] : 0.0f;
    __syncthreads();
    if (means) {   // the frame sums come from fe_generic_mean_kernel (one LANE a frame there: a sequential sum of frame_len samples by one
        // lane holds a whole workgroup here for 8 k cycles -- measured 1 ms of 6 at 32 kHz)
        if (tid < nfr) smean[tid] = means[(int64_t)blockIdx.y * num_frames + f0 + tid];
    } else if (tid < nfr) {   // pipeline.rs:115-116: raw_frame.iter().sum() / frame_len, samples scaled by 32768 first
        const float* fr = sp + tid * hop;
        float sum = 0.0f;
        int j = 0;
        for (; j + 16 <= frame_len; j += 16) {   // sixteen samples requested before the (dependent) additions: an LDS round trip per
            float v[16];                         // sample made this loop 38 us of a workgroup's 40
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = fr[j + u];
#pragma unroll
            for (int u = 0; u < 16; ++u) sum = sum + v[u] * 32768.0f;
        }
        for (; j < frame_len; ++j) sum = sum + fr[j] * 32768.0f;
        smean[tid] = sum / (float)frame_len;
    }
    __syncthreads();
    for (int i = tid; i < fpb * n; i += 256) {   // fe_generic_frame_kernel's value, at its bit-reversed slot
        const int f = i >> log2n, j = i & (n - 1);
        float v = 0.0f;
        if (j < frame_len && f < nfr) {
            const float* fr = sp + f * hop;
            const float m = smean[f];
            const float cur = fe::fsub(fe::fmul(fr[j], 32768.0f), m);
            float y = cur;
            if (j >= 1) y = fe::fsub(cur, fe::fmul(0.97f, fe::fsub(fe::fmul(fr[j - 1], 32768.0f), m)));  // pipeline.rs:140-142
            v = fe::fmul(y, window[j]);
        }
        const int r = (int)(__brev((unsigned)j) >> (32 - log2n));
        re[FE_PAD(f * n + r)] = v;
        im[FE_PAD(f * n + r)] = 0.0f;
    }
    __syncthreads();
    // The butterfly network, up to THREE stages per trip through LDS: a thread takes the 2^R points that stages s0 .. s0 + R - 1 combine
    // among themselves (indices base + j h, h = 2^s0), runs those stages on them in registers -- every butterfly with the twiddle and
    // the form (scalar below half = 4, fma from there: fft.rs:172-266) it has in the stage-by-stage network, so the same bits -- and
    // writes them back: 4 barriers for 1024 points instead of 10, 0.4 x the LDS traffic.
    for (int s0 = 0; s0 < log2n;) {
        const int R = log2n - s0 >= 3 ? 3 : log2n - s0, h = 1 << s0, gsz = 1 << R, groups = n >> R;
        for (int idx = tid; idx < fpb * groups; idx += 256) {
            const int f = idx / groups, g = idx - f * groups;
            const int lo = g & (h - 1), base = f * n + ((g >> s0) << (s0 + R)) + lo;
            float xr[8], xi[8];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < gsz) xr[j] = re[FE_PAD(base + j * h)], xi[j] = im[FE_PAD(base + j * h)];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                if (r >= R) break;
                const int half = h << r, off = half - 1;   // the stage's twiddles start at 1 + 2 + ... + half / 2
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (j >= gsz || (j >> r) & 1) continue;
                    const int k = lo + (j & ((1 << r) - 1)) * h;
                    const float wr = tw_re_g[off + k], wi = tw_im_g[off + k];   // (L1-resident: a copy in LDS costs occupancy)
                    if (half >= 4)
                        fe::bfly_fma(wr, wi, xr[j], xi[j], xr[j + (1 << r)], xi[j + (1 << r)]);
                    else
                        fe::bfly_scalar(wr, wi, xr[j], xi[j], xr[j + (1 << r)], xi[j + (1 << r)]);
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < gsz) re[FE_PAD(base + j * h)] = xr[j], im[FE_PAD(base + j * h)] = xi[j];
        }
        s0 += R;
        __syncthreads();
    }
    for (int i = tid; i < fpb * nb; i += 256) {   // fft_frames_kernel mode 1: bins 0 and n / 2 have im forced to 0 (fft.rs:256-261)
        const int f = i / nb, k = i - f * nb;
        const float r = re[FE_PAD(f * n + k)];
        const float q = (k == 0 || k == nb - 1) ? 0.0f : im[FE_PAD(f * n + k)];
        pw[i] = fe::power(r, q);
    }
    __syncthreads();
    for (int i = tid; i < nfr * n_mels; i += 256) {   // fe_generic_mel_kernel
        const int f = i / n_mels, m = i - f * n_mels;
        const float* pp = pw + f * nb + mstart[m];
        const float* wp = mw + moff[m];
        const int cnt = moff[m + 1] - moff[m];
        float acc = 0.0f;
        int t = 0;
        for (; t + 8 <= cnt; t += 8) {   // eight weights and eight powers requested before the (sequential) sum: a filter has up to ~60 taps,
            float w8[8], p8[8];          // and one memory round trip a tap was most of a workgroup's life (62 % of the wave cycles waiting)
#pragma unroll
            for (int u = 0; u < 8; ++u) w8[u] = wp[t + u], p8[u] = pp[t + u];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fe::fadd(acc, fe::fmul(w8[u], p8[u]));  // mel.rs:92-104
        }
        for (; t < cnt; ++t) acc = fe::fadd(acc, fe::fmul(wp[t], pp[t]));
        logmel[((int64_t)blockIdx.y * num_frames + f0 + f) * n_mels + m] = __logf(fmaxf(acc, 1e-5f));      // mel.rs:124-128
    }
}
__global__ void fe_generic_lfr_kernel(const float* __restrict__ x, int64_t t, int64_t d, int64_t m, int64_t n,
                                      int64_t t_lfr, float* __restrict__ out) {  // lfr.rs:18-54; grid.y = utterance
    const int64_t d_out = d * m, total = t_lfr * d_out, pad = (m - 1) / 2;
    x += (int64_t)blockIdx.y * t * d;
    out += (int64_t)blockIdx.y * total;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = idx / d_out, rem = idx - i * d_out, block = rem / d, k = rem - block * d;
        int64_t raw = i * n + block - pad;
        raw = raw < 0 ? 0 : (raw > t - 1 ? t - 1 : raw);
        out[idx] = x[raw * d + k];
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// Host side: mirrors SenseVoiceFrontend::new (pipeline.rs:38-65)
// ------------------------------------------------------------------------------------------------------
struct LeleFrontend {
    LeleCtx* ctx = nullptr;
    LeleFeatureConfig cfg{};
    int64_t frame_len = 0, hop_len = 0, n_fft = 0;
    bool fast = false;
    bool std_mel = false;  // mel bank has the default round structure (fully unrolled kernel variant)
    int dpp_mode = 0;  // 0: __shfl, 1: DPP row_ror (selected after a self-test)
    bool fused = true;  // frame sums computed inside fe_main_kernel (LELE_HIP_FE_FUSED=0 selects the two-kernel form)
    // generic path tables (configs other than 400/160/512)
    const float* g_window = nullptr;
    const int* g_mstart = nullptr;
    const int* g_moff = nullptr;
    const float* g_mw = nullptr;
    FeDev dev{};
    std::vector<void*> allocs;
    // optional per-kernel stopwatch (bench.py roofline block): 3 events per run, read back lazily
    bool profiling = false;
    std::vector<hipEvent_t> events;  // triples (before sum, between, after main)
    size_t events_used = 0;
};

namespace {

static const float PI_F = 3.14159265358979323846264338327950288f;

void host_hann(int64_t size, std::vector<float>& w) {  // window.rs:2-13
    w.resize(size);
    if (size == 1) w[0] = 1.0f;
    for (int64_t n = 0; size > 1 && n < size; ++n) w[n] = 0.5f * (1.0f - cosf(2.0f * PI_F * (float)n / (float)(size - 1)));
}

void host_twiddles(int64_t n, std::vector<float>& re, std::vector<float>& im) {  // kernels/fft.rs:136-157
    re.clear();
    im.clear();
    for (int64_t size = 2; size <= n; size *= 2) {
        int64_t half = size / 2, step = n / size;
        for (int64_t k = 0; k < half; ++k) {
            float angle = -2.0f * PI_F * (float)(k * step) / (float)n;
            re.push_back(cosf(angle));
            im.push_back(sinf(angle));
        }
    }
}

float hz_to_mel(float hz) { return 2595.0f * log10f(1.0f + hz / 700.0f); }       // mel.rs:1-3
float mel_to_hz(float mel) { return 700.0f * (powf(10.0f, mel / 2595.0f) - 1.0f); }  // mel.rs:4-6

// mel_filterbank (mel.rs:7-56): the dense [n_mels][n_fft / 2 + 1] triangles, HTK mel scale
void host_dense_mel(float sr, int64_t n_fft, int64_t n_mels, float f_min, float f_max, float* bank) {
    const int64_t n_freqs = n_fft / 2 + 1;
    const float mel_min = hz_to_mel(f_min), mel_max = hz_to_mel(f_max);
    const float mel_step = (mel_max - mel_min) / (float)(n_mels + 1);
    std::vector<float> hz(n_mels + 2), ff(n_freqs);
    for (int64_t i = 0; i < n_mels + 2; ++i) hz[i] = mel_to_hz(mel_min + (float)i * mel_step);
    for (int64_t i = 0; i < n_freqs; ++i) ff[i] = (float)i * sr / (float)n_fft;
    for (int64_t i = 0; i < n_mels; ++i) {
        const float fl = hz[i], fc = hz[i + 1], fr = hz[i + 2];
        for (int64_t j = 0; j < n_freqs; ++j) {
            const float f = ff[j];
            float val = 0.0f;
            if (f > fl && f < fc)
                val = (f - fl) / (fc - fl);
            else if (f >= fc && f < fr)
                val = (fr - f) / (fr - fc);
            bank[i * n_freqs + j] = val;
        }
    }
}

// SparseMelBank::new (mel.rs:58-90) over it: per filter (start_bin, weights[])
void host_sparse_mel(float sr, int64_t n_fft, int64_t n_mels, float f_min, std::vector<int>& start,
                     std::vector<std::vector<float>>& w) {
    const int64_t n_freqs = n_fft / 2 + 1;
    std::vector<float> bank((size_t)n_mels * n_freqs);
    host_dense_mel(sr, n_fft, n_mels, f_min, sr / 2.0f, bank.data());
    start.assign(n_mels, 0);
    w.assign(n_mels, {});
    for (int64_t i = 0; i < n_mels; ++i) {
        const float* row = bank.data() + i * n_freqs;
        int64_t s = 0;
        while (s < n_freqs && row[s] == 0.0f) ++s;
        int64_t e = n_freqs;
        while (e > s && row[e - 1] == 0.0f) --e;
        if (s < e) {
            start[i] = (int)s;
            w[i].assign(row + s, row + e);
        }
    }
}

template <typename T>
int upload(LeleFrontend* fe, const std::vector<T>& v, const T** out) {
    void* d = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);
    LELE_HIP_CHECK(hipMalloc(&d, bytes));
    fe->allocs.push_back(d);
    if (!v.empty()) LELE_HIP_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)d;
    return 0;
}

}  // namespace

extern "C" {

/* The table builders of lele::features as the library computes them itself (window.rs:2-13, mel.rs:7-56): host arithmetic, host
 * pointers, no ctx.  A binding exposes `hann_window` / `mel_filterbank` through these instead of restating the formulas. */
int lele_hip_hann_window(int64_t size, float* out) {
    LELE_REQUIRE(size >= 0 && (out || size == 0), "hann_window: bad argument");
    std::vector<float> w;
    host_hann(size, w);
    if (size) memcpy(out, w.data(), (size_t)size * 4);
    return 0;
}
float lele_hip_hz_to_mel_htk(float hz) { return hz_to_mel(hz); }      /* mel.rs:1-3 */
float lele_hip_mel_to_hz_htk(float mel) { return mel_to_hz(mel); }    /* mel.rs:4-6 */
int lele_hip_mel_filterbank(float sample_rate, int64_t n_fft, int64_t n_mels, float f_min, int32_t has_f_max, float f_max, float* out) {
    LELE_REQUIRE(n_fft >= 0 && n_mels >= 0 && (out || n_mels == 0), "mel_filterbank: bad argument");
    if (n_mels) host_dense_mel(sample_rate, n_fft, n_mels, f_min, has_f_max ? f_max : sample_rate / 2.0f, out);
    return 0;
}


int lele_hip_frontend_create(LeleCtx* ctx, const LeleFeatureConfig* cfg, LeleFrontend** out) {
    LELE_REQUIRE(ctx && cfg && out, "frontend_create: NULL argument");
    LELE_REQUIRE(cfg->sample_rate > 0 && cfg->n_mels > 0 && cfg->lfr_m > 0 && cfg->lfr_n > 0,
                 "frontend_create: invalid FeatureConfig");
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LeleFrontend* fe = new LeleFrontend();
    fe->ctx = ctx;
    fe->cfg = *cfg;
    fe->frame_len = (int64_t)((float)cfg->sample_rate * cfg->frame_length_ms / 1000.0f);  // pipeline.rs:39
    fe->n_fft = fe->frame_len > 400 ? 1024 : 512;                                        // pipeline.rs:40
    fe->hop_len = (int64_t)((float)cfg->sample_rate * cfg->frame_shift_ms / 1000.0f);     // pipeline.rs:42
    fe->fast = (fe->frame_len == fe::kFrame && fe->hop_len == fe::kHop && fe->n_fft == fe::kNfft &&
                cfg->n_mels <= 16 * kMaxMelRounds);
    if (!fe->fast) {
        // generic path: window + flat sparse mel bank; the FFT tables live in the ctx (features_ops.hip)
        if (fe->frame_len < 1 || fe->hop_len < 1 || fe->frame_len > fe->n_fft) {
            const long long fl = (long long)fe->frame_len, hl = (long long)fe->hop_len, nf = (long long)fe->n_fft;
            delete fe;
            // upstream indexes frame_buf[..fft_len] with frame_len (pipeline.rs:145-166) and divides by hop_len (:73)
            set_error("frontend_create: frame_len=%lld hop_len=%lld do not fit fft_len=%lld (the reference panics here)", fl, hl, nf);
            return 3;
        }
        std::vector<float> gwin;
        host_hann(fe->frame_len, gwin);
        std::vector<int> gstart;
        std::vector<std::vector<float>> gw;
        host_sparse_mel((float)cfg->sample_rate, fe->n_fft, cfg->n_mels, 20.0f, gstart, gw);  // pipeline.rs:45-51
        std::vector<int> goff(cfg->n_mels + 1, 0);
        std::vector<float> gflat;
        for (int64_t m = 0; m < cfg->n_mels; ++m) {
            goff[m] = (int)gflat.size();
            gflat.insert(gflat.end(), gw[m].begin(), gw[m].end());
        }
        goff[cfg->n_mels] = (int)gflat.size();
        int rc = 0;
        rc |= upload(fe, gwin, &fe->g_window);
        rc |= upload(fe, gstart, &fe->g_mstart);
        rc |= upload(fe, goff, &fe->g_moff);
        rc |= upload(fe, gflat, &fe->g_mw);
        if (rc) {
            lele_hip_frontend_destroy(fe);
            return rc;
        }
        *out = fe;
        return 0;
    }
    std::vector<float> win, twr, twi;
    host_hann(fe->frame_len, win);
    host_twiddles(fe->n_fft, twr, twi);
    // phase_a_fast folds table[0], table[1] == (1, -0) away: make sure this libm agrees (it must: cos(-0)=1)
    LELE_REQUIRE(twr[0] == 1.0f && twi[0] == 0.0f && twr[1] == 1.0f && twi[1] == 0.0f,
                 "frontend_create: unexpected twiddle table head");
    std::vector<float2> tw(512, make_float2(0.f, 0.f));
    for (size_t i = 0; i < twr.size(); ++i) tw[i] = make_float2(twr[i], twi[i]);
    twr.resize(512, 0.0f);
    twi.resize(512, 0.0f);
    std::vector<int> mstart;
    std::vector<std::vector<float>> mw;
    host_sparse_mel((float)cfg->sample_rate, fe->n_fft, cfg->n_mels, 20.0f, mstart, mw);  // pipeline.rs:45-51
    FeDev& d = fe->dev;
    d.n_mels = (int)cfg->n_mels;
    d.lfr_m = (int)cfg->lfr_m;
    d.lfr_n = (int)cfg->lfr_n;
    d.mel_rounds = (int)((cfg->n_mels + 15) / 16);
    std::vector<int> start_pad(d.mel_rounds * 16, 0);
    std::vector<float> melw;
    std::vector<int> step_off(d.mel_rounds + 1, 0);
    int off = 0;
    for (int rd = 0; rd < d.mel_rounds; ++rd) {
        size_t maxlen = 0;
        for (int pm = 0; pm < 16; ++pm) {
            int m = rd * 16 + pm;
            if (m < cfg->n_mels) {
                start_pad[m] = mstart[m];
                maxlen = std::max(maxlen, mw[m].size());
            }
        }
        step_off[rd] = off;
        for (size_t t = 0; t < maxlen; ++t)
            for (int pm = 0; pm < 16; ++pm) {
                int m = rd * 16 + pm;
                melw.push_back((m < cfg->n_mels && t < mw[m].size()) ? mw[m][t] : 0.0f);
            }
        off += (int)maxlen;
    }
    step_off[d.mel_rounds] = off;
    d.mel_steps = off;
    fe->std_mel = (d.mel_rounds == 5 && cfg->n_mels == 80);
    for (int rd = 0; fe->std_mel && rd < 5; ++rd) {
        if (step_off[rd] != kStdMelOff[rd] || step_off[rd + 1] != kStdMelOff[rd + 1]) fe->std_mel = false;
        for (int pm = 0; pm < 16; ++pm)
            if (start_pad[rd * 16 + pm] + kStdMelLen[rd] - 1 > 256) fe->std_mel = false;
    }
    if (lab_env("LELE_HIP_FE_GENERIC_MEL")) fe->std_mel = false;
    int rc = 0;
    rc |= upload(fe, twr, &d.tw_re);
    rc |= upload(fe, twi, &d.tw_im);
    rc |= upload(fe, tw, &d.tw);
    rc |= upload(fe, win, &d.window);
    rc |= upload(fe, melw, &d.melw);
    rc |= upload(fe, start_pad, &d.mel_start);
    rc |= upload(fe, step_off, &d.mel_step_off);
    if (rc) {
        lele_hip_frontend_destroy(fe);
        return rc;
    }
    const char* env = lab_env("LELE_HIP_FE_DPP");
    fe->dpp_mode = env ? atoi(env) : 1;  // 1: DPP row_ror (default), 0: __shfl (ds_bpermute)
    const char* envf = lab_env("LELE_HIP_FE_FUSED");
    fe->fused = envf ? atoi(envf) != 0 : true;
    *out = fe;
    return 0;
}

int lele_hip_frontend_destroy(LeleFrontend* fe) {
    if (!fe) return 0;
    (void)hipStreamSynchronize(fe->ctx->stream);
    for (void* p : fe->allocs) (void)hipFree(p);
    for (hipEvent_t e : fe->events) (void)hipEventDestroy(e);
    delete fe;
    return 0;
}

int lele_hip_frontend_out_rows(const LeleFrontend* fe, int64_t pcm_len, int64_t* rows, int64_t* cols,
                               int64_t* num_frames) {
    LELE_REQUIRE(fe, "frontend_out_rows: fe is NULL");
    int64_t nf = 0, t = 0;
    if (pcm_len >= fe->frame_len) {  // pipeline.rs:70-73
        nf = (pcm_len - fe->frame_len) / fe->hop_len + 1;
        t = (nf + fe->cfg.lfr_n - 1) / fe->cfg.lfr_n;  // lfr.rs:33
    }
    if (rows) *rows = t;
    if (cols) *cols = fe->cfg.n_mels * fe->cfg.lfr_m;
    if (num_frames) *num_frames = nf;
    return 0;
}

static int ilog2_i(int n) {
    int l = 0;
    while ((1 << (l + 1)) <= n) ++l;
    return l;
}
static int fe_run(LeleFrontend* fe, const LeleTensor* pcm, int64_t batch, int64_t pcm_len, LeleBuf* out,
                  bool want_logmel, int64_t* out_shape, int32_t* out_rank) {
    LeleCtx* ctx = fe->ctx;
    LELE_HIP_CHECK(hipSetDevice(ctx->device));
    LELE_REQUIRE(pcm->dtype == LELE_F32, "frontend: pcm must be f32");
    int64_t t_lfr = 0, cols = 0, nf = 0;
    lele_hip_frontend_out_rows(fe, pcm_len, &t_lfr, &cols, &nf);
    if (nf == 0 || batch == 0) {  // TensorView::empty()
        out->bytes = 0;
        if (out_rank) *out_rank = 0;
        return 0;
    }
    LELE_TRY(ctx->arena_reset());
    const void* dpcm = nullptr;
    LELE_TRY(ctx->dev_ptr(pcm, &dpcm));
    void* dmean = nullptr;
    LELE_TRY(ctx->get_scratch((size_t)batch * nf * sizeof(float), &dmean));
    const size_t out_elems = want_logmel ? (size_t)batch * nf * fe->cfg.n_mels : (size_t)batch * t_lfr * cols;
    LELE_TRY(out->reserve(out_elems * sizeof(float)));
    if (!fe->fast) {
        // generic path: as many utterances per pass as fit 256 MiB of scratch (means, frames [rows, n_fft], power [rows, bins], logmel
        // [rows, mels]; rows = utterances x frames) -- five launches a pass (looping over utterances on the host made the launches
        // the cost: 256 x 30 s at 16 kHz / 20 ms frames 16.1 ms)
        const int64_t bins = fe->n_fft / 2 + 1, nm = fe->cfg.n_mels;
        // the fused form (fe_generic_fused_kernel): frames per workgroup so that their LDS image stays under 64 KB; anything larger
        // (n_fft > 4096) keeps the four kernels
        if (fe->fused && fe->n_fft <= 4096 && fe->frame_len <= fe->n_fft && batch <= 65535) {   // (lab: LELE_HIP_FE_FUSED=0 keeps the four kernels)
            const int n = (int)fe->n_fft;
            // (the kernel is bound by latency under its barriers, so by occupancy against work per workgroup: 4 frames of 1024 points with
            // the twiddles in LDS = 55 KB = two workgroups a CU ran 15 ms where the four kernels took 12)
            int fpb = n <= 512 ? 4 : n <= 1024 ? 2 : 1;   // measured (tools/fe_fpb.sh): n = 512: 3.9 / 2.6 / 2.4 / 3.1 ms for 1 / 2 / 4 / 8; n = 1024: 6.1 / 5.2 / 7.1 / 17.2
            if (const char* e = lab_env("LELE_HIP_FE_FPB")) fpb = std::max(1, atoi(e));   // (lab) frames per workgroup
            const size_t padded = (size_t)fpb * n + (((size_t)fpb * n) >> 5) + 1;
            const size_t lds = (2 * padded + (size_t)fpb * bins + (size_t)(fpb - 1) * fe->hop_len + fe->frame_len + fpb) * 4;
            if (lds <= 96 * 1024) {
                const float *twr = nullptr, *twi = nullptr;
                LELE_TRY(fft_twiddles(ctx, n, &twr, &twi));
                LELE_HIP_CHECK(ensure_dyn_lds(reinterpret_cast<const void*>(fe_generic_fused_kernel), (int)lds));
                const int64_t per_utt_lm = nf * nm * 4;
                const int64_t ubf = want_logmel ? batch : std::max<int64_t>(1, std::min<int64_t>(batch, (int64_t(1) << 28) / std::max<int64_t>(per_utt_lm, 1)));
                void* glm = nullptr;
                if (!want_logmel) LELE_TRY(ctx->arena_alloc((size_t)ubf * nf * nm * 4, &glm));
                for (int64_t u = 0; u < batch; u += ubf) {
                    const int64_t nu = std::min<int64_t>(ubf, batch - u);
                    const float* up = (const float*)dpcm + u * pcm_len;
                    float* lm = want_logmel ? (float*)out->data + u * nf * nm : (float*)glm;
                    hipLaunchKernelGGL(fe_generic_mean_kernel, dim3((unsigned)((nu * nf + 63) / 64)), dim3(64), 0, ctx->stream, up, nu * nf, nf, pcm_len,
                                       (int)fe->frame_len, (int)fe->hop_len, (float*)dmean);
                    hipLaunchKernelGGL(fe_generic_fused_kernel, dim3((unsigned)((nf + fpb - 1) / fpb), (unsigned)nu), dim3(256), lds, ctx->stream, up,
                                       pcm_len, nf, (int)fe->frame_len, (int)fe->hop_len, n, ilog2_i(n), fpb, fe->g_window, twr, twi, (int)nm,
                                       fe->g_mstart, fe->g_moff, fe->g_mw, (const float*)dmean, lm);
                    if (!want_logmel) {
                        dim3 lg((unsigned)std::max<int64_t>(1, std::min<int64_t>((t_lfr * cols + 255) / 256, 8192)), (unsigned)nu);
                        hipLaunchKernelGGL(fe_generic_lfr_kernel, lg, dim3(256), 0, ctx->stream, (const float*)lm, nf, nm, fe->cfg.lfr_m,
                                           fe->cfg.lfr_n, t_lfr, (float*)out->data + u * t_lfr * cols);
                    }
                }
                LELE_HIP_CHECK(hipGetLastError());
                if (want_logmel) {
                    if (batch == 1) return set_shape(out_shape, out_rank, {nf, nm});
                    return set_shape(out_shape, out_rank, {batch, nf, nm});
                }
                return 0;
            }
        }
        const int64_t per_utt = nf * (4 + fe->n_fft * 4 + bins * 4 + nm * 4);
        const int64_t ub = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(batch, 65535), (int64_t(1) << 28) / std::max<int64_t>(per_utt, 1)));
        void *gm = nullptr, *gf = nullptr, *gp = nullptr, *gl = nullptr;
        LELE_TRY(ctx->arena_alloc((size_t)ub * nf * 4, &gm));
        LELE_TRY(ctx->arena_alloc((size_t)ub * nf * fe->n_fft * 4, &gf));
        LELE_TRY(ctx->arena_alloc((size_t)ub * nf * bins * 4, &gp));
        if (!want_logmel) LELE_TRY(ctx->arena_alloc((size_t)ub * nf * nm * 4, &gl));
        auto blocks = [](int64_t n) { return dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((n + 255) / 256, 8192))); };
        for (int64_t u = 0; u < batch; u += ub) {
            const int64_t nu = std::min<int64_t>(ub, batch - u), rows = nu * nf;
            const float* up = (const float*)dpcm + u * pcm_len;
            float* lm = want_logmel ? (float*)out->data + u * nf * nm : (float*)gl;
            hipLaunchKernelGGL(fe_generic_mean_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(64), 0, ctx->stream, up, rows, nf, pcm_len,
                               (int)fe->frame_len, (int)fe->hop_len, (float*)gm);
            hipLaunchKernelGGL(fe_generic_frame_kernel, blocks(rows * fe->n_fft), dim3(256), 0, ctx->stream, up, (const float*)gm,
                               fe->g_window, rows, nf, pcm_len, (int)fe->frame_len, (int)fe->hop_len, (int)fe->n_fft, (float*)gf);
            LELE_TRY(fft_rows_power(ctx, (const float*)gf, rows, fe->n_fft, (float*)gp));
            hipLaunchKernelGGL(fe_generic_mel_kernel, blocks(rows * nm), dim3(256), 0, ctx->stream, (const float*)gp, rows, (int)bins,
                               (int)nm, fe->g_mstart, fe->g_moff, fe->g_mw, lm);
            if (!want_logmel) {
                dim3 lg = blocks(t_lfr * cols);
                lg.y = (unsigned)nu;
                hipLaunchKernelGGL(fe_generic_lfr_kernel, lg, dim3(256), 0, ctx->stream, (const float*)lm, nf, nm,
                                   fe->cfg.lfr_m, fe->cfg.lfr_n, t_lfr, (float*)out->data + u * t_lfr * cols);
            }
        }
        LELE_HIP_CHECK(hipGetLastError());
        if (want_logmel) {
            if (batch == 1) return set_shape(out_shape, out_rank, {nf, nm});
            return set_shape(out_shape, out_rank, {batch, nf, nm});
        }
        return 0;
    }
    const int aligned16 = ((uintptr_t)dpcm % 16 == 0) && (pcm_len % 4 == 0);
    dim3 grid((unsigned)((nf + 63) / 64), (unsigned)batch);
    hipEvent_t* ev = nullptr;
    if (fe->profiling) {
        if (fe->events_used + 3 > fe->events.size()) {
            for (int i = 0; i < 3; ++i) {
                hipEvent_t e;
                LELE_HIP_CHECK(hipEventCreate(&e));
                fe->events.push_back(e);
            }
        }
        ev = &fe->events[fe->events_used];
        fe->events_used += 3;
        LELE_HIP_CHECK(hipEventRecord(ev[0], ctx->stream));
    }
    if (!fe->fused)
        hipLaunchKernelGGL(fe_frame_sum_kernel, grid, dim3(64), 0, ctx->stream, (const float*)dpcm, pcm_len, pcm_len, nf,
                           (float*)dmean, aligned16);
    if (ev) LELE_HIP_CHECK(hipEventRecord(ev[1], ctx->stream));
    const size_t mel_lds = (size_t)fe->dev.mel_steps * 16 * sizeof(float);
    float* o = want_logmel ? nullptr : (float*)out->data;
    float* lm = want_logmel ? (float*)out->data : nullptr;
#define FE_LAUNCH(MODE, STD, FUS)                                                                                     \
    hipLaunchKernelGGL((fe_main_kernel<MODE, STD, FUS>), grid, dim3(256), mel_lds, ctx->stream, (const float*)dpcm, \
                       pcm_len, nf, t_lfr, (const float*)dmean, fe->dev, o, lm, aligned16)
#define FE_LAUNCH2(MODE, STD) \
    do {                      \
        if (fe->fused)        \
            FE_LAUNCH(MODE, STD, true); \
        else                  \
            FE_LAUNCH(MODE, STD, false); \
    } while (0)
    if (fe->dpp_mode == 1) {
        if (fe->std_mel) FE_LAUNCH2(1, true); else FE_LAUNCH2(1, false);
    } else {
        if (fe->std_mel) FE_LAUNCH2(0, true); else FE_LAUNCH2(0, false);
    }
#undef FE_LAUNCH2
#undef FE_LAUNCH
    LELE_HIP_CHECK(hipGetLastError());
    if (ev) LELE_HIP_CHECK(hipEventRecord(ev[2], ctx->stream));
    if (want_logmel) {
        if (batch == 1) return set_shape(out_shape, out_rank, {nf, fe->cfg.n_mels});
        return set_shape(out_shape, out_rank, {batch, nf, fe->cfg.n_mels});
    }
    return 0;
}

int lele_hip_frontend_compute(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                              int32_t* out_rank) {
    LELE_REQUIRE(fe && pcm && out, "frontend_compute: NULL argument");
    const int64_t n = numel(pcm);
    LELE_TRY(fe_run(fe, pcm, 1, n, out, false, out_shape, out_rank));
    if (out->bytes == 0) return 0;
    int64_t t = 0, cols = 0;
    lele_hip_frontend_out_rows(fe, n, &t, &cols, nullptr);
    return set_shape(out_shape, out_rank, {t, cols});
}

int lele_hip_frontend_compute_batch(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                                    int32_t* out_rank) {
    LELE_REQUIRE(fe && pcm && out, "frontend_compute_batch: NULL argument");
    LELE_REQUIRE(pcm->rank == 2, "frontend_compute_batch: pcm must be [batch, pcm_len]");
    const int64_t batch = pcm->shape[0], n = pcm->shape[1];
    LELE_TRY(fe_run(fe, pcm, batch, n, out, false, out_shape, out_rank));
    if (out->bytes == 0) return 0;
    int64_t t = 0, cols = 0;
    lele_hip_frontend_out_rows(fe, n, &t, &cols, nullptr);
    return set_shape(out_shape, out_rank, {batch, t, cols});
}

int lele_hip_frontend_logmel(LeleFrontend* fe, const LeleTensor* pcm, LeleBuf* out, int64_t* out_shape,
                             int32_t* out_rank) {
    LELE_REQUIRE(fe && pcm && out, "frontend_logmel: NULL argument");
    if (pcm->rank == 2) return fe_run(fe, pcm, pcm->shape[0], pcm->shape[1], out, true, out_shape, out_rank);
    return fe_run(fe, pcm, 1, numel(pcm), out, true, out_shape, out_rank);
}

int lele_hip_frontend_set_profiling(LeleFrontend* fe, int on) {
    LELE_REQUIRE(fe, "frontend_set_profiling: fe is NULL");
    fe->profiling = on != 0;
    fe->events_used = 0;
    return 0;
}

int lele_hip_frontend_profile_read(LeleFrontend* fe, float* sum_kernel_ms, float* main_kernel_ms, int64_t* runs) {
    LELE_REQUIRE(fe, "frontend_profile_read: fe is NULL");
    LELE_HIP_CHECK(hipStreamSynchronize(fe->ctx->stream));
    double a = 0, b = 0;
    const size_t n = fe->events_used / 3;
    for (size_t i = 0; i < n; ++i) {
        float m0 = 0, m1 = 0;
        LELE_HIP_CHECK(hipEventElapsedTime(&m0, fe->events[3 * i], fe->events[3 * i + 1]));
        LELE_HIP_CHECK(hipEventElapsedTime(&m1, fe->events[3 * i + 1], fe->events[3 * i + 2]));
        a += m0;
        b += m1;
    }
    if (sum_kernel_ms) *sum_kernel_ms = n ? (float)(a / n) : 0.f;
    if (main_kernel_ms) *main_kernel_ms = n ? (float)(b / n) : 0.f;
    if (runs) *runs = (int64_t)n;
    fe->events_used = 0;
    return 0;
}

}  // extern "C"
