// adaface_amd — shared device/host definitions for the gfx950 (MI355X) kernels.
//
// Storage type T is either __bf16 (throughput mode) or float (parity mode).
// All activations are NHWC ("token-major"): [B, H*W, C] with C contiguous, so
// the reference's 'b c h w -> b (h w) c' rearranges (attention.py:327,335) are
// no-ops and every 1x1 conv / Linear is a plain row-major GEMM.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define AF_WAVE 64

#define HIP_CHECK_RET(expr)                                                       \
  do {                                                                            \
    hipError_t _e = (expr);                                                       \
    if (_e != hipSuccess) {                                                       \
      af_set_error_msg("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),     \
                       __FILE__, __LINE__);                                       \
      return -2;                                                                  \
    }                                                                             \
  } while (0)

void af_set_error_msg(const char* fmt, ...);

// hipFuncAttributeMaxDynamicSharedMemorySize is a property of (kernel, device): remember per device id where it has
// been raised (one mask per kernel instantiation; a handle may live on any GPU of the node)
// (distinct handles may be driven from distinct host threads: the mask is read and updated atomically; setting the attribute
// twice is harmless)
static inline int af_ensure_dynamic_lds(unsigned long long& done_mask, const void* fn, int bytes) {
  int dev = 0;
  HIP_CHECK_RET(hipGetDevice(&dev));
  if (dev >= 0 && dev < 64 && ((__atomic_load_n(&done_mask, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return 0;
  HIP_CHECK_RET(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  if (dev >= 0 && dev < 64) __atomic_fetch_or(&done_mask, 1ull << dev, __ATOMIC_RELEASE);
  return 0;
}

// per-kernel-class HIP-event profiling (enabled only by bench.py; see af_prof_* in adaface_hip.h)
// classes 5-7 split the eight-wave ping-pong kernel out of the conv/linear class by instantiation, so that the bench can
// quote ONE kernel (conv_gemm_pp_kernel<160, true>, the 3x3 convolutions) with its own launch count and duration
enum { AF_K_CONV_GEMM = 0, AF_K_ATTENTION = 1, AF_K_GROUPNORM = 2, AF_K_LAYERNORM = 3, AF_K_OTHER = 4,
       AF_K_PP160_GATHER = 5, AF_K_PP160_PLAIN = 6, AF_K_PP128 = 7, AF_K_PP_FP8 = 8, AF_K_HALO8 = 9, AF_K_COUNT = 10 };
// (process-wide diagnostics, updated by every launch: atomics, since distinct handles may be used from distinct host threads)
extern int g_af_prof_enabled;
extern int g_af_prof_stride;              // bracket only every stride-th launch of a class (>= 1)
extern std::atomic<long> g_af_prof_seen[AF_K_COUNT];   // launches seen per class since af_prof_reset
extern std::atomic<double> g_af_flops_issued;          // FLOPs handed to GEMM / conv / attention launches since af_flops_issued(reset): what the
                                          // path EXECUTES (phase-decomposed upsamplers and the shared CFG prefix spend fewer
                                          // than the reference's algorithm; bench.py reports both fractions)
void af_prof_begin_impl(int cls, hipStream_t s, double flops, double bytes);
void af_prof_end_impl(hipStream_t s);
struct AfProfScope {
  hipStream_t s;
  bool on;
  AfProfScope(int cls, hipStream_t s_, double flops, double bytes) : s(s_), on(((g_af_prof_enabled >> cls) & 1) != 0) {
    if (flops != 0.0) {
      double cur = g_af_flops_issued.load(std::memory_order_relaxed);
      while (!g_af_flops_issued.compare_exchange_weak(cur, cur + flops, std::memory_order_relaxed)) {}
    }
    // an event pair costs ~9 us of stream time on MI355X (measured: bracketing every GEMM and attention launch slowed
    // the 50-step batch by 10 %), so the bench samples every stride-th launch of a class instead of all of them
    if (on) on = (g_af_prof_seen[cls].fetch_add(1, std::memory_order_relaxed) % g_af_prof_stride) == 0;
    if (on) af_prof_begin_impl(cls, s, flops, bytes);
  }
  ~AfProfScope() {
    if (on) af_prof_end_impl(s);
  }
};

// Tuning / diagnostic knobs (planner thresholds, forced tiles).  One plain struct the launchers read: it is filled ONCE
// from the AF_* environment variables when the library is loaded and afterwards changed only through af_knob_set
// (tests force a kernel variant that way).  Nothing on the launch path calls getenv, and no knob changes results.
struct AfKnobs {
  int splitk_target;        // AF_SPLITK_TARGET        four-wave kernels: workgroups aimed for when slicing K
  int conv_halo;            // AF_CONV_HALO            0 = never use the LDS-halo 3x3 kernel
  int gemm_pp;              // AF_GEMM_PP              0 = never use the eight-wave ping-pong kernel
  int gemm_pp_minfill;      // AF_GEMM_PP_MINFILL      ping-pong kernel from this grid fill (percent)
  int gemm_tile;            // AF_GEMM_TILE            >= 0: force a four-wave tile
  int gemm_splitk;          // AF_GEMM_SPLITK          >= 1: force the number of K slices
  int gemm_groupm;          // AF_GEMM_GROUPM          >= 1: force the grouped tile order
  int gemm_dma;             // AF_GEMM_DMA             0 / 1: force register / LDS-DMA staging in the four-wave kernel
  int pp_direct;            // AF_PP_DIRECT            0 / 1: force the LDS / direct epilogue of the ping-pong kernel
  int attn_ring;            // AF_ATTN_RING            bit 0: dh-40, bit 1: dh-80 (>= 256 keys; bit 2: any key count) bf16 attention on the eight-wave ring kernel
  int gn_small;             // AF_GN_SMALL             0 = no single-launch GroupNorm for small maps
  int conv_tap_inner;       // AF_CONV_TAP_INNER       0 = ping-pong convs walk K tap-outermost (the round-1 order)
  int ln_fuse;              // AF_LN_FUSE              0 = stand-alone LayerNorm kernels in front of the transformer GEMMs
  int geglu_rowpanel;       // AF_GEGLU_ROWPANEL       row-panel kernel (K = 320, >= 32768 rows): 0 = never, 1 = GEGLU only, 2 = the plain
                            //                         GEMMs of the 64x64-level transformers too, 3 = and its K = 640 form (GEGLU and
                            //                         q / k / v of the 32x32 level, >= 16384 rows), 4 = and the K = 1280 form
                            //                         ([>= 4096, 1280] -> 1280 of the 16x16 level)
  int conv_halo8;           // AF_CONV_HALO8           bit 0: 3x3 / stride-1 convs of the 64x64 / 32x32 / 16x16 maps on conv3x3_halo8_kernel,
                            //                         bit 1: those of the 8x8 maps on conv3x3_s8_kernel; 0 = all on the gathering kernel
  int conv_fast_taps;       // AF_CONV_FAST_TAPS       0 = ping-pong convs recompute every tap's bounds check in the staging phase
  int pp_stagger;           // AF_PP_STAGGER           merged schedule: 1 = the two wave groups issue their LDS-DMA pieces behind alternate MFMAs
  int gn_producer;          // AF_GN_PRODUCER          0 = GroupNorm always runs its own statistics pass (no sums from the producer convolution)
  int conv_up_phase4;       // AF_CONV_UP_PHASE4       0 = upsampled 3x3 convolutions gather all nine taps from the upsampled map
  int pp_sched;             // AF_PP_SCHED             eight-wave kernel: 0 = round-1 compute phase (two K halves, a full LDS drain
                            //                         after each), 1 = block-ordered compute phase, 2 = merged (no staging phase)
  int attn_short;           // AF_ATTN_SHORT           0 = cross-attention (<= 96 keys) stays on the flash kernels
  int gemm_m128;            // AF_GEMM_M128            0 = no 128 x 160 tile GEMM for the few-row plain GEMMs (16x16 level)
  int small_m_tile64;       // AF_SMALL_M_TILE64       0 = GEMMs with <= 1024 rows and K <= 2048 keep the cost model's tile / K slices
  int gn_consumer;          // AF_GN_CONSUMER          0 = the SpatialTransformer's GroupNorm always runs its own apply pass (never
                            //                         in the prologue of a row-panel proj_in, 64x64 level)
  int xattn_fused;          // AF_XATTN_FUSED          0 = cross-attention of the 64x64-level transformers as three launches (to_q,
                            //                         short-key attention, to_out + residual) instead of xattn_fused_kernel
  int plan_log;             // AF_PLAN_LOG             1 = one stderr line per GEMM / convolution launch: shape, tile, K slices (lab)
};
extern AfKnobs g_af_knobs;

// ---------------------------------------------------------------------------
// scalar conversion helpers
// ---------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16>(bf16 v) { return (float)v; }

template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 16-byte vector of storage elements (8 bf16 or 4 float)
template <typename T> struct Vec16 {
  static constexpr int N = 16 / sizeof(T);
  union {
    uint4 u;
    T e[N];
  };
};

// 4 consecutive storage elements (8 B for bf16, 16 B for float)
template <typename T> struct Quad;
template <> struct Quad<bf16> {
  union {
    uint2 u;
    bf16 e[4];
  };
  __device__ __forceinline__ void load(const bf16* p) { u = *reinterpret_cast<const uint2*>(p); }
  __device__ __forceinline__ void store(bf16* p) const { *reinterpret_cast<uint2*>(p) = u; }
};
template <> struct Quad<float> {
  union {
    uint4 u;
    float e[4];
  };
  __device__ __forceinline__ void load(const float* p) { u = *reinterpret_cast<const uint4*>(p); }
  __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<uint4*>(p) = u; }
};

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
// exact (erf) GELU, as F.gelu default (attention.py:41)
// erf by Abramowitz & Stegun 7.1.26 (|abs err| <= 1.5e-7, i.e. fp32 rounding level): one v_rcp, one v_exp and a
// degree-5 Horner chain instead of libm's branchy erff — the GEGLU epilogue evaluates it 8192 times per tile and
// was the dominant cost of the K=320 feed-forward projections.
__device__ __forceinline__ float erf_as_f(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float poly = fmaf(1.061405429f, t, -1.453152027f);
  poly = fmaf(poly, t, 1.421413741f);
  poly = fmaf(poly, t, -0.284496736f);
  poly = fmaf(poly, t, 0.254829592f);
  poly *= t;
  const float e = __builtin_amdgcn_exp2f(-1.44269504088896340736f * ax * ax);
  const float r = fmaf(-poly, e, 1.0f);
  return copysignf(r, x);
}
__device__ __forceinline__ float gelu_erf_f(float x) {
  return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f));
}
// GELU for bf16 OUTPUTS: x * Phi(x) with Phi(x) ~ sigmoid(x (a + b x^2)), (a, b) = (1.5974834, 0.0706872) fitted to the
// exact erf form: |error| <= 4.0e-4 for every x (largest near |x| = 2.8, where one bf16 step of the result is 1.6e-2;
// the small-|x| series agrees to 1e-3 of the quadratic term).  4 plain VALU + v_exp + v_rcp instead of 14 + 2: the
// GEGLU epilogue of the ping-pong kernel evaluates it 16384 times per tile and is VALU-bound at K = 320.  The f32
// (parity) kernels keep gelu_erf_f.
__device__ __forceinline__ float gelu_bf16out_f(float x) {
  const float u = x * fmaf(x * x, -0.07068715223f * 1.44269504089f, -1.59748341624f * 1.44269504089f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(u));
}

// The same for PAIRS of values, without transcendentals: Phi(x) ~ 0.5 + xc q(xc^2), xc = x clamped to +-4, q a degree-6
// minimax-weighted fit (max |error| of x Phi(x): 1.9e-4 over all x).  Written on 2-wide vectors so that hipcc emits the packed
// f32 instructions (v_pk_mul_f32 / v_pk_fma_f32: two values per instruction at full rate): 9 packed + 2 v_med3 per PAIR
// instead of 2 x (6 plain + v_exp + v_rcp, the last two quarter-rate) -- the GEGLU epilogue is VALU-bound (half of the
// row-panel kernel's time, scripts/lab/ablate_conv.sh geglu).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 gelu_bf16out_f2(f32x2 x) {
  f32x2 xc;
  xc.x = __builtin_amdgcn_fmed3f(x.x, -4.0f, 4.0f);
  xc.y = __builtin_amdgcn_fmed3f(x.y, -4.0f, 4.0f);
  const f32x2 s = xc * xc;
  f32x2 q = s * 2.27794120e-08f + (-1.59850396e-06f);
  q = q * s + 4.79536935e-05f;
  q = q * s + (-8.13999317e-04f);
  q = q * s + 8.77231965e-03f;
  q = q * s + (-6.45729896e-02f);
  q = q * s + 3.97883296e-01f;
  const f32x2 ph = xc * q + 0.5f;
  return x * ph;
}

// ---------------------------------------------------------------------------
// MFMA wrapper: one "fragment step" consumes a 16-byte fragment per lane from
// each operand (= 32 bytes of K per row: 16 bf16 or 8 float).
//   bf16 : one v_mfma_f32_32x32x16_bf16      (lane l: row l&31, k = 8*(l>>5)+j)
//   float: four v_mfma_f32_32x32x2_f32, MFMA j taking element j of the
//          fragment, i.e. k = 4*(l>>5)+j : a fixed permutation of K applied
//          identically to both operands, so the sum is unchanged.
// D[i][j] = sum_k A[i][k]*B[k][j]; D col = lane&31, row = (r&3)+8*(r>>2)+4*(lane>>5).
// ---------------------------------------------------------------------------
template <typename T> struct Mma;
template <> struct Mma<bf16> {
  static __device__ __forceinline__ void step(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  static __device__ __forceinline__ void step(const uint4& a, const uint4& b, f32x16& c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x),
                                             __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y),
                                             __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z),
                                             __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w),
                                             __builtin_bit_cast(float, b.w), c, 0, 0, 0);
  }
};

// row of the 32x32 accumulator held in register r of lane-half h
__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// ---------------------------------------------------------------------------
// kernel parameter blocks (plain structs passed by value)
// ---------------------------------------------------------------------------
enum { AF_EPI_NONE = 0, AF_EPI_GEGLU = 1 };

// Implicit-GEMM convolution / linear:  out[m][n] = sum_k X(m,k) * W[n][k] (+epilogue)
//   m = (b, oy, ox), k = (ky, kx, c).  ks==1 degenerates to a row-major GEMM.
struct ConvGemmParams {
  const void* src;        // activations, T, NHWC
  long src_batch_stride;  // elements between samples
  int ldc;                // elements between pixels (>= Cin)
  int Cin;                // input channels used (multiple of BK)
  int Hs, Ws;             // stored source spatial dims
  int up;                 // 1: nearest-2x upsample folded into the gather
  int Hi, Wi;             // logical input dims (Hs<<up, Ws<<up)
  int Ho, Wo;             // output dims
  int ks, stride, pad;
  const void* W;          // T, [Wrows][ldw]
  int ldw;                // elements between weight rows (>= K)
  int Wrows;              // rows that may be read (padded rows are zero)
  int M, N, K;            // N = valid GEMM columns (multiple of 4)
  int k_logical;          // un-padded K for algorithmic FLOP accounting (0 = K)
  const float* bias;      // [N] or null
  const void* rowbias;    // T, [B][ldrb] per-sample bias (time embedding) or null
  int ldrb;
  const void* residual;   // T, [M][ldr] or null
  int ldr;
  void* out;              // T, [M][ldo]
  int ldo;
  int epilogue;
  float alpha;            // scale applied to the accumulator before bias
  // batched GEMM (blockIdx.z): element strides
  long bs_src, bs_w, bs_out, bs_res;
  // split-K (set by the launcher): K slices write fp32 slabs ws[splitk][M][N]
  int splitk;
  void* ws;
  int group_m;            // grouped tile ordering (set by the launcher): M tiles swept per N tile
  int howo_shift, wo_shift;  // log2(Ho*Wo), log2(Wo) when they are powers of two, else -1 (set by the launcher)
  // LayerNorm folded into the ping-pong GEMM (bf16, af_model.hip run_xfmr):
  //   producer side: ln_stats_out != null -> besides its output the GEMM writes, per output row m and per 80-column slab
  //     part = 2 * (n tile) + wave group, the sum and the sum of squares of the bf16-ROUNDED values it stored:
  //     ln_stats_out[(part * M + m) * 2 + {0, 1}]  (fp32; fixed summation order, no atomics: deterministic)
  //     (ln_finalize_kernel turns them into ln_stats[m] = (mu, rstd): one tiny launch instead of every column tile's
  //     workgroup re-reducing the slabs in front of its pipeline — that cost 20-60 us per consumer launch)
  //   consumer side: ln_stats != null -> the A operand is the un-normalised x and W holds W * gamma; the epilogue applies
  //     out = rstd[m] * (acc - mu[m] * ln_colsum[n]) + bias[n], bias = W beta + b precomputed at load time.  GEGLU
  //     composes with it.
  float* ln_stats_out;
  const float* ln_stats;
  const float* ln_colsum;
  // consumer on a ROW-PANEL kernel (a workgroup owns its rows for all of N): ln_parts_n > 0 -> ln_stats holds the producer's
  // partial sums [ln_parts_n][M][2] and the kernel finalises (mu, rstd) of its rows itself: mu = sum * ln_inv_count, ...
  int ln_parts_n;
  float ln_inv_count, ln_eps;
  int k_tap_inner;        // ping-pong kernel (set by the launcher): K walked as (channel chunk, tap) instead of (tap, chunk)
  int pp_epilogue;        // ping-pong kernel (set by the launcher): 0 = direct for GEGLU / split-K slabs and LDS
                          // otherwise, 1 = always through LDS, 2 = always direct
  // fp8 (OCP e4m3) operands on the block-scaled MFMA (ping-pong kernel only; bf16 in every other respect: bias,
  // residual, output).  src = fp8 NHWC (ldc, src_batch_stride in BYTES), W = fp8 [Wrows][ldw] with K laid out in 64-channel
  // UNITS, unit u = (channel chunk u / taps, tap u % taps), zero-padded to a multiple of 128 = K; Cin = real channel
  // count (multiple of 64).  Scales are powers of two applied by the MFMA itself (E8M0): w_scale[n] per output channel,
  // x_scale_e8 for the whole activation tensor (the producer multiplied by 2^(127 - x_scale_e8)).
  int pp_stagger;         // merged schedule (set by the launcher): the wave groups stage behind alternate MFMAs
  int fast_taps;          // ping-pong kernel (set by the launcher): per-piece tap validity masks instead of per-tap bounds arithmetic
  int fp8;
  const unsigned char* w_scale;
  int x_scale_e8;
  // Nearest-2x upsample + 3x3 convolution as FOUR 2x2 convolutions on the stored (low-resolution) map, one per output
  // phase (dy, dx) = (Y & 1, X & 1): the three upsampled rows a 3x3 window covers are only two stored rows, so the taps that
  // read the same stored pixel are summed ONCE at load time (af_launch_up_phase4_weights) and the GEMM does 4/9 of the work.
  //   caller: W_up4 = phase weights [4][Wrows][4 * Cin] (K = (ty, tx, c)) next to the ordinary 3x3 weights; the launcher
  //   switches when the shape qualifies (bf16, up = 1, ks = 3, stride 1, power-of-two maps, no residual / time bias)
  //   kernel (set by the launcher): phase4 = 1 -> blockIdx.y = phase, padding (1 - dy, 1 - dx), rows m = (b, y, x) of the
  //   STORED map, output pixel (2 y + dy, 2 x + dx) of the [B][2 Hs][2 Ws] map
  const void* W_up4;
  int phase4;
  // GroupNorm statistics from the producer (eight-wave kernels, bf16 outputs, one K slice): besides its output the
  // convolution writes, per 64-row slab of a sample and per group of gn_cpg channels, the sum and the sum of squares of
  // the bf16-ROUNDED values it stored, in the layout gn_apply_kernel folds: gn_stats_out[((b * npart + slab) * 32 + group) * 2],
  // npart = Ho * Wo / 64 (fixed order, no atomics).  The consumer GroupNorm then needs no statistics pass over the tensor.
  float* gn_stats_out;
  int gn_cpg;
  // GroupNorm applied by the CONSUMER (row-panel kernels only: their activation rows sit in registers for the whole launch):
  // src holds the UN-normalised x, gn_ab [B][2][K] the per-sample scale / shift (af_launch_groupnorm_fold), gn_hw the rows
  // per sample (a multiple of the panel height); the prologue replaces x by bf16(fma(x, a, b)) -- the value gn_apply_kernel
  // would have stored
  const float* gn_ab;
  int gn_hw;
};

struct AfGemmPlan {
  int tile;         // 0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64
  int splitk;       // >= 1
  size_t ws_bytes;  // fp32 slab workspace needed when splitk > 1
  int halo_tw;      // 0 = implicit-GEMM kernel; 16 / 32 = LDS-halo 3x3 kernel with that patch width
  int group_m;      // grouped tile ordering: M tiles swept per N tile
};

struct AttnParams {
  const void* q; const void* k; const void* v; void* o;  // T
  int ldq, ldk, ldv, ldo;        // row strides (elements)
  long bsq, bsk, bsv, bso;       // batch strides (elements)
  int Nq, Nk, H;
  float scale;
  float* lse;   // optional [B][H][Nq]: log2-domain log-sum-exp of the scaled scores (m + log2 l); null = off
  int causal;   // != 0: query i attends to keys <= i only (the CLIP text tower's mask, modeling_clip causal mask)
  const void* vt_pack;   // bf16, optional: V packed as resident MFMA fragments (af_launch_attn_short_pack) -> the short-key
                         // cross-attention kernel runs when Nk <= 96 and dh is 40 or 80; null = the flash kernels
};
