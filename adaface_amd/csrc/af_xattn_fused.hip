// adaface_amd — ONE kernel per cross-attention layer of the 64x64-level transformers (bf16, C = 320, 8 heads x 40, <= 80
// context keys):
//
//     x = x + attn2(norm2(x), context)        BasicTransformerBlock._forward, /root/reference/ldm/modules/attention.py:275-285
//       q = to_q(LayerNorm(x))                CrossAttention.forward, attention.py:172-196 (to_q has no bias)
//       o = softmax(q k^T * scale) v          attention.py:197-243 (k, v: the hoisted, step-invariant context projections)
//       x + to_out(o)                         attention.py:245-257 (Linear + Dropout(0))
//
// Round 3 ran this as three launches that each stream the [65536, 320] tensor: the LayerNorm-folded q projection (row-panel
// GEMM, 27.7 us), the register-resident short-key attention (30.8 us) and to_out + residual (27.7 us).  Here a workgroup owns
// 256 rows (eight waves x 32 rows, all inside one sample) for ALL heads and x is read once / the result written once:
//
//   phase 1  q^T = Wq' x^T            the wave's 32 rows stay in 80 VGPRs as MFMA B fragments (as rowpanel_kernel), Wq * gamma
//                                     streams through a five-slot LDS ring in 80-column tiles (one head PAIR per tile); the
//                                     LayerNorm fold rstd (acc - mu colsum) + W beta is applied on the accumulators and the
//                                     result is rounded to bf16 exactly where the unfused path rounds it (its q tensor).
//                                     The accumulator of a 16x16x32 MFMA holds, per lane, 4 consecutive channels of one
//                                     token -- two such blocks ARE a B-operand fragment (token on the lane, 8 K values), for
//                                     any fixed assignment of channels to K slots: no LDS, no shuffles.
//   phase 2  per head: S^T = K q^T    K pre-scaled by scale * log2(e) and PACKED per layer at af_set_context in exactly that
//            softmax, O^T = V^T P^T   K-slot order (pack_kv_kernel); the packs of a head pair (38 KB) are staged by LDS-DMA
//                                     into one of two buffers while the previous pair computes.  S^T accumulators -> P^T operand
//                                     and O^T accumulators -> operand of phase 3 by the same trick; the softmax denominator is
//                                     an MFMA with an all-ones A operand (sums the bf16-rounded P the numerator uses).
//                                     O overwrites q in the same 80 registers head by head.
//   phase 3  out^T = Wo' O^T          Wo with its K dimension permuted (at weight load) into the order in which the O^T
//            + bias + x               accumulator blocks come back as operands; residual rows re-read (L2 / MALL), LayerNorm
//                                     partial sums of the stored values for norm3's consumer, 16-byte row stores through a
//                                     wave-private transposition tile.
//
// Per workgroup 1184 MFMAs per wave against ~40 KB of x + out traffic; the three launches it replaces moved 5 x that.
// f32 (parity) mode, other widths / head sizes, conv attention (needs the log-sum-exp) and > 80 keys keep the unfused path.
#include "af_kernels.h"

#include <type_traits>

namespace xf {
typedef bf16 T;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

constexpr int C = 320, H = 8, DH = 40, KC = 5, NPAIR = 4;
constexpr int BNT = 80, NIW = 5, NTILE = 4;          // 80-column tiles = one head pair; 16-column blocks per tile
constexpr int MJ = 2, WROWS = 32, BM = 256;
constexpr int D = 4, NS = 5;                         // weight ring: prefetch distance (steps), slots
constexpr int WBYTES = BNT * 128, WPIECES = BNT / 8; // 10 KiB per slot, 10 pieces of 1 KiB
constexpr int NKB = 5, SMAX = 16 * NKB;              // key blocks of 16 (keys >= Nk masked)
constexpr int NSP = (NKB + 1) / 2;                   // 32-key k steps of PV
constexpr int K_FRAGS = 2 * NKB * 2, V_FRAGS = 2 * 3 * NSP;          // 1-KiB fragments per head pair: 20 + 18
constexpr int KV_PPW = (K_FRAGS + V_FRAGS + 7) / 8;  // LDS-DMA pieces per wave and pair (5)
constexpr int KVBUF = KV_PPW * 8 * 1024;             // 40 KiB
constexpr int RING0 = 0, KV0 = NS * WBYTES, VEC0 = KV0 + 2 * KVBUF;         // 51200 + 81920 = 133120
constexpr int LDS_BYTES = VEC0 + 3 * C * 4;          // + the column sums / folded bias of to_q and the bias of to_out (fp32)
constexpr int OPITCH = BNT * 2 + 16, OBYTES = WROWS * OPITCH;               // per-wave output transposition tile (5632 B)
constexpr int CPR = BNT / 8, NST = WROWS * CPR / 64;                        // 10 chunks per row segment, 5 stores per lane
static_assert(8 * OBYTES <= 2 * KVBUF && (WROWS * CPR) % 64 == 0 && LDS_BYTES <= 160 * 1024, "LDS plan");

// packed K / V of one (sample, head pair): KV_PPW * 8 fragments of [64 lanes][8 bf16]
//   K fragment (hh, kb, s)  at ((hh * NKB + kb) * 2 + s):   lane (key = 16 kb + (lane & 15), g = lane >> 4), element j
//   V fragment (hh, blk, sp) at K_FRAGS + (hh * 3 + blk) * NSP + sp: lane (row = lane & 15 of block blk, g), element j
constexpr long PACK_ELEMS_PER_PAIR = (long)KV_PPW * 8 * 512;

// channel d (0..39) of head hh (0 = first head of the pair) held by K slot (s, g, j) of the S^T contraction, or -1 = zero.
// Phase 1 leaves q as five 16-channel accumulator blocks per pair (block t: pair columns 16 t .. 16 t + 15; a lane holds
// rows 4 g + e).  Head 0 = columns 0..39, head 1 = columns 40..79.  k step s = 0 takes blocks (0, 1) / (3, 4), s = 1 takes
// block 2 in BOTH halves of the fragment (its second copy multiplies zeros).
__host__ __device__ inline int k_slot_channel(int hh, int s, int g, int j) {
  const int r = 4 * g + (j & 3);
  if (s == 0) {
    const int blk = (hh == 0 ? 0 : 3) + (j >> 2);
    const int col = 16 * blk + r;
    return col - 40 * hh;
  }
  if (j >= 4) return -1;
  const int col = 32 + r;                            // block 2
  return (hh == 0) ? (col < 40 ? col : -1) : (col >= 40 ? col - 40 : -1);
}
// channel d held by row r of O^T block blk (0, 1: own blocks; 2: the block the two heads of a pair share) of head hh, or -1
__host__ __device__ inline int v_row_channel(int hh, int blk, int r) {
  if (blk < 2) return 16 * blk + r;
  return hh == 0 ? (r < 8 ? 32 + r : -1) : (r >= 8 ? 32 + (r - 8) : -1);
}
// key held by K slot (g, j) of PV k step sp
__host__ __device__ inline int pv_slot_key(int sp, int g, int j) { return 32 * sp + 16 * (j >> 2) + 4 * g + (j & 3); }
// input channel (0..319) of to_out held at position pos (0..63) of K chunk kc of the permuted weight: phase 2 leaves O as
// 20 blocks (pair p: blocks 5 p .. 5 p + 4 = head 0 d 0-15, head 0 d 16-31, shared, head 1 d 0-15, head 1 d 16-31)
__host__ __device__ inline int wo_pos_channel(int kc, int pos) {
  const int u = pos >> 5, g = (pos >> 3) & 3, j = pos & 7;
  const int bl = 4 * kc + 2 * u + (j >> 2), r = 4 * g + (j & 3);
  const int pair = bl / 5, t = bl % 5;
  const int base = 80 * pair;
  switch (t) {
    case 0: return base + r;
    case 1: return base + 16 + r;
    case 2: return r < 8 ? base + 32 + r : base + 40 + 32 + (r - 8);
    case 3: return base + 40 + r;
    default: return base + 56 + r;
  }
}

__global__ __launch_bounds__(256) void pack_kv_kernel(const bf16* __restrict__ kv, int ldk, long bsk, int Nk, int B, float sl2,
                                                      bf16* __restrict__ pack) {
  const long total = (long)B * NPAIR * PACK_ELEMS_PER_PAIR;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int j = (int)(r & 7); r >>= 3;
    const int lane = (int)(r & 63); r >>= 6;
    const int frag = (int)(r % (KV_PPW * 8)); r /= (KV_PPW * 8);
    const int pair = (int)(r % NPAIR);
    const int b = (int)(r / NPAIR);
    const int l15 = lane & 15, g = lane >> 4;
    float val = 0.f;
    if (frag < K_FRAGS) {
      const int s = frag & 1, kb = (frag >> 1) % NKB, hh = (frag >> 1) / NKB;
      const int key = 16 * kb + l15, d = k_slot_channel(hh, s, g, j);
      if (key < Nk && d >= 0) val = (float)kv[(long)b * bsk + (long)key * ldk + (2 * pair + hh) * DH + d] * sl2;
    } else if (frag < K_FRAGS + V_FRAGS) {
      const int f = frag - K_FRAGS;
      const int sp = f % NSP, blk = (f / NSP) % 3, hh = f / (3 * NSP);
      const int key = pv_slot_key(sp, g, j), d = v_row_channel(hh, blk, l15);
      if (key < Nk && d >= 0) val = (float)kv[(long)b * bsk + (long)key * ldk + C + (2 * pair + hh) * DH + d];
    }
    pack[i] = (bf16)val;
  }
}

// to_out weight [C rows][ldw] -> the same rows with K permuted (wo_pos_channel)
__global__ __launch_bounds__(256) void permute_wo_kernel(const bf16* __restrict__ w, int ldw, int rows, bf16* __restrict__ wp, int ldp) {
  const long total = (long)rows * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / C), k = (int)(i - (long)n * C);
    wp[(long)n * ldp + k] = w[(long)n * ldw + wo_pos_channel(k >> 6, k & 63)];
  }
}

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}
template <int OFF> __device__ __forceinline__ u32x4 lds_read128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// the wait names the registers it covers ("+v"): no consumer of them can be scheduled above it
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm2(u32x4& a, u32x4& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}
__device__ __forceinline__ void wait_lgkm0() {
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
__device__ __forceinline__ f32x4 mma(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  bf16x2 v;
  v.x = (bf16)a;
  v.y = (bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ u32x2 pack4(const f32x4& v) { return u32x2{pack2(v[0], v[1]), pack2(v[2], v[3])}; }
__device__ __forceinline__ u32x4 cat(const u32x2& a, const u32x2& b) { return u32x4{a.x, a.y, b.x, b.y}; }

struct Params {
  const void* x; int ldx;                // [M][ldx] bf16: the un-normalised rows (A operand of q) and the residual
  int M;
  const float* ln_stats;                 // (mu, rstd) per row, or with ln_parts_n > 0 the producer's partial sums [parts][M][2]
  int ln_parts_n; float ln_inv_count, ln_eps;
  const void* wq; int ldwq;              // to_q weight * gamma [C][ldwq]
  const float* q_colsum; const float* q_bias;   // column sums of wq, W beta (+ bias)
  const void* kvpack; int rows_per_sample;      // [B][NPAIR][PACK_ELEMS_PER_PAIR]
  const void* wo; int ldwo;              // to_out weight, K permuted [C][ldwo]
  const float* o_bias;
  void* out; int ldo;
  float* ln_stats_out;                   // [4][M][2] partial sums of the stored rows (parts 0 / 2 = columns 0-159 / 160-319), or null
  int Nk;
};

__global__ __launch_bounds__(512) void xattn_fused_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.x * BM, r0 = m0 + wid * WROWS;
  const int bsample = m0 / p.rows_per_sample;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.x)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.wq)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_wo = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(reinterpret_cast<const T*>(p.wo)), 0, (int)0xFFFFFFF0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_kv = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<T*>(reinterpret_cast<const T*>(p.kvpack) + (long)bsample * NPAIR * PACK_ELEMS_PER_PAIR), 0,
      (int)(NPAIR * PACK_ELEMS_PER_PAIR * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<T*>(p.out), 0, (int)0xFFFFFFF0u, 0x00020000);

  // ---- K / V packs of a head pair -> buffer (pair & 1): KV_PPW pieces per wave, lane-linear ----
  auto stage_kv = [&](int pair) {
    char* base = smem + KV0 + (pair & 1) * KVBUF;
#pragma unroll
    for (int q = 0; q < KV_PPW; ++q) {
      const int piece = wid + 8 * q;
      lds_dma16(rs_kv, base + piece * 1024, (unsigned)(lane * 16), (unsigned)(pair * (int)(PACK_ELEMS_PER_PAIR * 2) + piece * 1024));
    }
  };
  stage_kv(0);
  stage_kv(1);
  // the per-column vectors of the two epilogues -> LDS (read there as 16-byte vectors, two blocks at a time: fetched from
  // memory in the epilogue, ten float4 per tile next to 80 row and 60 q registers, they spilled)
  if (tid < 3 * (C / 4)) {
    const int v = tid / (C / 4), c4 = tid - v * (C / 4);
    const float* src = v == 0 ? p.q_colsum : (v == 1 ? p.q_bias : p.o_bias);
    const float4 val = src ? *reinterpret_cast<const float4*>(src + 4 * c4) : float4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<float4*>(smem + VEC0 + (v * C + 4 * c4) * 4) = val;
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (visible to the other waves after the first step's barrier)

  // ---- the wave's 32 rows as MFMA B fragments: block j, K chunk kc, half u -> k = 64 kc + 32 u + 8 g ----
  u32x4 xr[MJ][KC][2];
  float ln_mu[MJ], ln_rs[MJ];
#pragma unroll
  for (int j = 0; j < MJ; ++j) {
    const int m = r0 + j * 16 + l15;
    const bool ok = m < p.M;
    const unsigned xo = ok ? (unsigned)((long)m * p.ldx * 2) + (unsigned)g * 16u : 0xFFFFFFFFu;
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
      for (int u = 0; u < 2; ++u)
        xr[j][kc][u] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo, (kc * 64 + u * 32) * 2, 0));
    float2 st = float2{0.f, 0.f};
    if (ok && p.ln_parts_n > 0) {      // (same order and arithmetic as ln_finalize_kernel / rowpanel_kernel)
      float s1 = 0.f, s2 = 0.f;
      for (int q = 0; q < p.ln_parts_n; ++q) {
        const float2 pq2 = *reinterpret_cast<const float2*>(p.ln_stats + ((long)q * p.M + m) * 2);
        s1 += pq2.x;
        s2 += pq2.y;
      }
      const float mu = s1 * p.ln_inv_count;
      const float var = fmaxf(s2 * p.ln_inv_count - mu * mu, 0.f);
      st = float2{mu, __builtin_amdgcn_rsqf(var + p.ln_eps)};
    } else if (ok) {
      st = *reinterpret_cast<const float2*>(p.ln_stats + (long)m * 2);
    }
    ln_mu[j] = st.x;
    ln_rs[j] = st.y;
  }

  // ---- weight ring: step t = (tile t / KC, chunk t % KC); t < 20: to_q, t >= 20: to_out ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  const int nwq = wid < WPIECES - 8 ? 2 : 1;         // pieces per wave and step (10 pieces over 8 waves)
  unsigned wq_off[2], wo_off[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    wq_off[q] = (unsigned)(((wid + 8 * q) * 8 + srow) * p.ldwq * 2) + dchunk * 16u;
    wo_off[q] = (unsigned)(((wid + 8 * q) * 8 + srow) * p.ldwo * 2) + dchunk * 16u;
  }
  const unsigned tstride_q = (unsigned)(BNT * p.ldwq * 2), tstride_o = (unsigned)(BNT * p.ldwo * 2);
  constexpr int T_ALL = 2 * NTILE * KC;
  auto stage = [&](int t, int slot) {
    const bool live = t < T_ALL;
    const bool is_o = t >= NTILE * KC;
    const int tt = is_o ? t - NTILE * KC : t;
    const int ntl = tt / KC, kc = tt - ntl * KC;
    const unsigned so = (unsigned)ntl * (is_o ? tstride_o : tstride_q) + (unsigned)kc * 128u;
    const __amdgpu_buffer_rsrc_t rs = is_o ? rs_wo : rs_wq;
    lds_dma16(rs, smem + RING0 + slot * WBYTES + wid * 1024, live ? (is_o ? wo_off[0] : wq_off[0]) : 0xFFFFFFFFu, so);
    if (wid < WPIECES - 8)
      lds_dma16(rs, smem + RING0 + slot * WBYTES + (wid + 8) * 1024, live ? (is_o ? wo_off[1] : wq_off[1]) : 0xFFFFFFFFu, so);
  };
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)(g ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)((g + 4) ^ (lane & 7)) * 16u;
  const unsigned w_base = (unsigned)(l15 * 128);
  const int cl = 4 * g;

  f32x4 acc[NIW][MJ];
  u32x4 wf[NIW][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < NIW; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };
  // one K step of a projection tile: the B operand of row block j is (b0[j], b1[j]); weight blocks read three ahead
  auto step = [&](int slot, int t, int stage_slot, const u32x4 (&b0)[MJ], const u32x4 (&b1)[MJ]) {
    const unsigned a0 = lds0 + (unsigned)(RING0 + slot * WBYTES) + w_base + fch0;
    const unsigned a1 = lds0 + (unsigned)(RING0 + slot * WBYTES) + w_base + fch1;
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wf[i][0] = lds_read128<i * 2048>(a0);
      wf[i][1] = lds_read128<i * 2048>(a1);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    rd(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NIW>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int issued = (i + 3) < NIW ? (i + 3) : NIW;
      wait_lgkm2<2 * (issued - i - 1)>(wf[i][0], wf[i][1]);
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        acc[i][j] = mma(wf[i][0], b0[j], acc[i][j]);
        acc[i][j] = mma(wf[i][1], b1[j], acc[i][j]);
        asm volatile("" : "+v"(acc[i][j]));
      }
      if constexpr (i + 3 < NIW) rd(std::integral_constant<int, i + 3>{});
      if constexpr (i == 1) stage(t + D, stage_slot);
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_lgkm0();
  };

#pragma unroll
  for (int i = 0; i < D; ++i) stage(i, i);
  int cur = 0;
  auto step_head = [&](int extra_stores) {
    // own pieces of this step landed: everything but the pieces of the next D - 1 steps (and an epilogue's younger stores)
    if (extra_stores) { if (nwq == 2) wait_vm<2 * (D - 1) + NST>(); else wait_vm<(D - 1) + NST>(); }
    else { if (nwq == 2) wait_vm<2 * (D - 1)>(); else wait_vm<(D - 1)>(); }
    __builtin_amdgcn_s_barrier();
  };

  // =========================== phase 1: q (bf16) as operand blocks qo[pair][block][row block] ===========================
  u32x2 qo[NPAIR][NIW][MJ];
  zero_acc();
  static_for<0, NTILE>([&](auto ntc) {
    constexpr int nt = decltype(ntc)::value;
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      step_head(0);
      const int prev = cur == 0 ? NS - 1 : cur - 1;
      const u32x4 b0[MJ] = {xr[0][kc][0], xr[1][kc][0]}, b1[MJ] = {xr[0][kc][1], xr[1][kc][1]};
      step(cur, nt * KC + kc, prev, b0, b1);
      cur = cur + 1 == NS ? 0 : cur + 1;
    });
    // LayerNorm fold on the accumulators: q = rstd (acc - mu colsum) + (W beta), rounded to bf16 as the q tensor was
    {
      const unsigned va = lds0 + (unsigned)VEC0 + 16u * (unsigned)g;
      u32x4 csv[NIW], bvv[NIW];
      static_for<0, NIW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        csv[i] = lds_read128<(nt * BNT + i * 16) * 4>(va);
        bvv[i] = lds_read128<(C + nt * BNT + i * 16) * 4>(va);
        if constexpr (i >= 1) {
          constexpr int k = i - 1;
          wait_lgkm2<2>(csv[k], bvv[k]);
          const f32x4 cs = __builtin_bit_cast(f32x4, csv[k]), bv = __builtin_bit_cast(f32x4, bvv[k]);
#pragma unroll
          for (int j = 0; j < MJ; ++j) {
            f32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = (acc[k][j][e] - ln_mu[j] * cs[e]) * ln_rs[j] + bv[e];
            qo[nt][k][j] = pack4(v);
            asm volatile("" : "+v"(qo[nt][k][j]));   // (defined HERE: hipcc otherwise sinks the fold + conversion to the use in phase 2 and keeps the fp32 accumulators alive)
            acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
        }
      });
      {
        constexpr int k = NIW - 1;
        wait_lgkm2<0>(csv[k], bvv[k]);
        const f32x4 cs = __builtin_bit_cast(f32x4, csv[k]), bv = __builtin_bit_cast(f32x4, bvv[k]);
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[k][j][e] - ln_mu[j] * cs[e]) * ln_rs[j] + bv[e];
          qo[nt][k][j] = pack4(v);
          asm volatile("" : "+v"(qo[nt][k][j]));
          acc[k][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
    }
  });

  // =========================== phase 2: attention, head pair by head pair; O replaces q in qo ===========================
  u32x4 ones;
  ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  asm volatile("" : "+v"(ones));
  const int Nk = p.Nk;
  const bool tail_mask = Nk < SMAX;
  static_for<0, NPAIR>([&](auto pc) {
    constexpr int pr = decltype(pc)::value;
    if constexpr (pr == 2) { wait_vm<KV_PPW>(); __builtin_amdgcn_s_barrier(); }   // own pieces of pair 2 landed (pair 3's may fly)
    if constexpr (pr == 3) { wait_vm<0>(); __builtin_amdgcn_s_barrier(); }
    const unsigned kvb = lds0 + (unsigned)(KV0 + (pr & 1) * KVBUF) + (unsigned)lane * 16u;
    f32x4 oc[MJ];                                   // the O^T block the two heads share (rows 0-7: head 0, rows 8-15: head 1)
    float inv0[MJ];
    static_for<0, 2>([&](auto hc) {
      constexpr int hh = decltype(hc)::value;
      constexpr int tb0 = hh == 0 ? 0 : 3;          // own q blocks of this head
      // ---- S^T = K q^T: NKB key blocks x 2 k steps, both row blocks of the wave ----
      f32x4 sc[MJ][NKB];
      {
        u32x4 qb0[MJ], qb1[MJ];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          qb0[j] = cat(qo[pr][tb0][j], qo[pr][tb0 + 1][j]);
          qb1[j] = cat(qo[pr][2][j], qo[pr][2][j]);
        }
        constexpr unsigned kbase = (unsigned)(hh * NKB * 2 * 1024);
        u32x4 kf[NKB][2];
        static_for<0, NKB>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          kf[kb][0] = lds_read128<kbase + (kb * 2 + 0) * 1024>(kvb);
          kf[kb][1] = lds_read128<kbase + (kb * 2 + 1) * 1024>(kvb);
        });
        static_for<0, NKB>([&](auto kbc) {
          constexpr int kb = decltype(kbc)::value;
          wait_lgkm2<2 * (NKB - 1 - kb)>(kf[kb][0], kf[kb][1]);
#pragma unroll
          for (int j = 0; j < MJ; ++j) {
            sc[j][kb] = mma(kf[kb][0], qb0[j], f32x4{0.f, 0.f, 0.f, 0.f});
            sc[j][kb] = mma(kf[kb][1], qb1[j], sc[j][kb]);
          }
        });
      }
      // ---- exact softmax over the keys < Nk of this lane's token (rows 4 g + e of every key block; the four lane groups of a
      // token meet through two cross-lane maxima) ----
      u32x4 pb[MJ][NSP];
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        if (tail_mask) {                 // (wave-uniform) only a partial or empty key block has rows to mask
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb)
            if (16 * kb + 16 > Nk) {
#pragma unroll
              for (int e = 0; e < 4; ++e)
                if (16 * kb + 4 * g + e >= Nk) sc[j][kb][e] = -INFINITY;
            }
        }
        float mx = fmaxf(fmaxf(sc[j][0][0], sc[j][0][1]), fmaxf(sc[j][0][2], sc[j][0][3]));
#pragma unroll
        for (int kb = 1; kb < NKB; ++kb) mx = fmaxf(fmaxf(mx, fmaxf(sc[j][kb][0], sc[j][kb][1])), fmaxf(sc[j][kb][2], sc[j][kb][3]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
#pragma unroll
        for (int sp = 0; sp < NSP; ++sp) {
          f32x4 e0, e1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            e0[e] = __builtin_amdgcn_exp2f(sc[j][2 * sp][e] - mx);
            e1[e] = (2 * sp + 1 < NKB) ? __builtin_amdgcn_exp2f(sc[j][2 * sp + 1 < NKB ? 2 * sp + 1 : 0][e] - mx) : 0.f;
          }
          pb[j][sp] = cat(pack4(e0), pack4(e1));
        }
      }
      // ---- O^T = V^T P^T (own blocks 0, 1 + the shared block) and the denominator (ones x P^T) ----
      f32x4 o0[MJ], o1[MJ], ls[MJ];
      {
        constexpr unsigned vbase = (unsigned)((K_FRAGS + hh * 3 * NSP) * 1024);
        u32x4 vf[3][NSP];
        static_for<0, 3 * NSP>([&](auto fc) {
          constexpr int f = decltype(fc)::value;
          vf[f / NSP][f % NSP] = lds_read128<vbase + f * 1024>(kvb);
        });
        static_for<0, NSP>([&](auto spc) {
          constexpr int sp = decltype(spc)::value;
#pragma unroll
          for (int j = 0; j < MJ; ++j) ls[j] = mma(ones, pb[j][sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : ls[j]);
        });
        static_for<0, NSP>([&](auto spc) {
          constexpr int sp = decltype(spc)::value;
          wait_lgkm<3 * NSP - 1 - sp>(vf[0][sp]);
#pragma unroll
          for (int j = 0; j < MJ; ++j) o0[j] = mma(vf[0][sp], pb[j][sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o0[j]);
        });
        static_for<0, NSP>([&](auto spc) {
          constexpr int sp = decltype(spc)::value;
          wait_lgkm<2 * NSP - 1 - sp>(vf[1][sp]);
#pragma unroll
          for (int j = 0; j < MJ; ++j) o1[j] = mma(vf[1][sp], pb[j][sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o1[j]);
        });
        static_for<0, NSP>([&](auto spc) {
          constexpr int sp = decltype(spc)::value;
          wait_lgkm<NSP - 1 - sp>(vf[2][sp]);
#pragma unroll
          for (int j = 0; j < MJ; ++j) oc[j] = mma(vf[2][sp], pb[j][sp], (hh == 0 && sp == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : oc[j]);
        });
      }
      // ---- normalise; O blocks take the place of the q blocks they came from ----
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const float inv = __builtin_amdgcn_rcpf(ls[j][0]);
        qo[pr][tb0][j] = pack4(o0[j] * inv);
        qo[pr][tb0 + 1][j] = pack4(o1[j] * inv);
        asm volatile("" : "+v"(qo[pr][tb0][j]), "+v"(qo[pr][tb0 + 1][j]));
        if constexpr (hh == 0) {
          inv0[j] = inv;
        } else {
          const float ic = g < 2 ? inv0[j] : inv;
          qo[pr][2][j] = pack4(oc[j] * ic);
          asm volatile("" : "+v"(qo[pr][2][j]));
        }
      }
    });
    if constexpr (pr < 2) {                          // refill this pair's buffer with pair + 2 once every wave has left it
      __builtin_amdgcn_s_barrier();
      stage_kv(pr + 2);
    }
  });

  // =========================== phase 3: out = to_out(O) + bias + x ===========================
  char* otile = smem + KV0 + wid * OBYTES;           // (the K / V buffers are dead: every wave passes the next barrier first)
  unsigned orow[NST];
#pragma unroll
  for (int i = 0; i < NST; ++i) {
    const int c = lane + 64 * i;
    const int row = c / CPR, ch = c - row * CPR;
    const int m = r0 + row;
    orow[i] = m < p.M ? (unsigned)((long)m * p.ldo * 2) + (unsigned)ch * 16u : 0xFFFFFFFFu;
  }
  float ps[MJ] = {0.f, 0.f}, pq[MJ] = {0.f, 0.f};
  int after_epi = 0;
  static_for<0, NTILE>([&](auto ntc) {
    constexpr int nt = decltype(ntc)::value;
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      step_head(after_epi > 0 ? 1 : 0);
      if (after_epi > 0) --after_epi;
      const int prev = cur == 0 ? NS - 1 : cur - 1;
      // K chunk kc = O blocks 4 kc .. 4 kc + 3 (of the 20 = 4 pairs x 5)
      constexpr int bA = 4 * kc, bB = 4 * kc + 1, bC = 4 * kc + 2, bD = 4 * kc + 3;
      const u32x4 b0[MJ] = {cat(qo[bA / 5][bA % 5][0], qo[bB / 5][bB % 5][0]), cat(qo[bA / 5][bA % 5][1], qo[bB / 5][bB % 5][1])};
      const u32x4 b1[MJ] = {cat(qo[bC / 5][bC % 5][0], qo[bD / 5][bD % 5][0]), cat(qo[bC / 5][bC % 5][1], qo[bD / 5][bD % 5][1])};
      step(cur, (NTILE + nt) * KC + kc, prev, b0, b1);
      cur = cur + 1 == NS ? 0 : cur + 1;
    });
    // ---- epilogue of the 80-column tile: + bias + residual, bf16, LayerNorm partial sums, transposition tile, row stores ----
    constexpr int ncol = nt * BNT;
    u32x4 bvec[NIW];
    {
      const unsigned va = lds0 + (unsigned)VEC0 + 16u * (unsigned)g;
      static_for<0, NIW>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        bvec[i] = lds_read128<(2 * C + ncol + i * 16) * 4>(va);
      });
      wait_lgkm0();
    }
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      const int m = r0 + j * 16 + l15;
      const bool mok = m < p.M;
      Quad<T> rq[NIW];
      const T* rp = reinterpret_cast<const T*>(p.x) + (long)(mok ? m : 0) * p.ldx + ncol + cl;
#pragma unroll
      for (int i = 0; i < NIW; ++i) rq[i].load(rp + i * 16);
#pragma unroll
      for (int i = 0; i < NIW; ++i) {
        const f32x4 bp = __builtin_bit_cast(f32x4, bvec[i]);
        Quad<T> o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float v = acc[i][j][e] + bp[e] + to_f32<T>(rq[i].e[e]);
          o.e[e] = from_f32<T>(v);
          const float vr = mok ? to_f32<T>(o.e[e]) : 0.f;     // the value the consumer will read
          ps[j] += vr;
          pq[j] += vr * vr;
        }
        o.store(reinterpret_cast<T*>(otile + (j * 16 + l15) * OPITCH) + i * 16 + cl);
        acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
    if constexpr (nt == 1 || nt == 3) {
      // per-row partial sums in the layout of the unfused to_out launch (row-panel, 160-column tiles): parts 0 / 2 hold the
      // sums of columns 0-159 / 160-319, parts 1 / 3 are zero
      if (p.ln_stats_out) {
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          const int m = r0 + j * 16 + l15;
          float s1 = ps[j], s2 = pq[j];
          s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
          s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
          if (m < p.M && g == 0) {
            *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(nt - 1) * p.M + m) * 2) = float2{s1, s2};
            *reinterpret_cast<float2*>(p.ln_stats_out + ((long)nt * p.M + m) * 2) = float2{0.f, 0.f};
          }
          ps[j] = pq[j] = 0.f;
        }
      }
    }
    // (wave-private tile: no barrier; all offsets and chunks first, then the stores back to back: a 16-byte buffer store with an
    // SGPR offset must not be followed by a VALU write of its data registers, scripts/check_isa_hazards.py)
    u32x4 chunk[NST];
#pragma unroll
    for (int i = 0; i < NST; ++i) {
      const int c = lane + 64 * i;
      const int row = c / CPR, ch = c - row * CPR;
      chunk[i] = *reinterpret_cast<const u32x4*>(otile + row * OPITCH + ch * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(chunk[i], rs_o, orow[i], ncol * 2, 0);
    __builtin_amdgcn_sched_barrier(0);
    after_epi = D;
  });
  wait_vm<0>();
}
}  // namespace xf

// ---------------------------------------------------------------------------------------------------------------------------
long af_xattn_fused_pack_elems(int B, int H, int dh, int Nk) {
  if (H != xf::H || dh != xf::DH || Nk <= 0 || Nk > xf::SMAX || B <= 0) return 0;
  return (long)B * xf::NPAIR * xf::PACK_ELEMS_PER_PAIR;
}
int af_launch_xattn_fused_pack(const void* kv, int ldk, long bsk, int Nk, int B, float scale, void* pack, hipStream_t stream) {
  const long total = (long)B * xf::NPAIR * xf::PACK_ELEMS_PER_PAIR;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(xf::pack_kv_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16*>(kv), ldk, bsk, Nk, B,
                     scale * 1.44269504088896340736f, reinterpret_cast<bf16*>(pack));
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
int af_launch_xattn_fused_permute_wo(const void* w, int ldw, int rows, void* wp, int ldp, hipStream_t stream) {
  hipLaunchKernelGGL(xf::permute_wo_kernel, dim3(400), dim3(256), 0, stream, reinterpret_cast<const bf16*>(w), ldw, rows,
                     reinterpret_cast<bf16*>(wp), ldp);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}
bool af_xattn_fused_ok(int M, int rows_per_sample, int C, int H, int dh, int Nk) {
  return C == xf::C && H == xf::H && dh == xf::DH && Nk > 0 && Nk <= xf::SMAX && rows_per_sample % xf::BM == 0 && M % xf::BM == 0 && M > 0;
}
std::atomic<long> g_af_xattn_fused_launches{0};
int af_launch_xattn_fused(const AfXattnFusedParams& a, hipStream_t stream) {
  if (!af_xattn_fused_ok(a.M, a.rows_per_sample, xf::C, xf::H, xf::DH, a.Nk) || a.ldx % 8 || a.ldo % 8 || a.ldwq % 8 || a.ldwo % 8) {
    af_set_error_msg("xattn_fused: unsupported shape M=%d rows/sample=%d Nk=%d", a.M, a.rows_per_sample, a.Nk);
    return -1;
  }
  static unsigned long long attr_done = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&xf::xattn_fused_kernel), xf::LDS_BYTES)) return rc;
  xf::Params p;
  p.x = a.x; p.ldx = a.ldx; p.M = a.M;
  p.ln_stats = a.ln_stats; p.ln_parts_n = a.ln_parts_n; p.ln_inv_count = a.ln_inv_count; p.ln_eps = a.ln_eps;
  p.wq = a.wq; p.ldwq = a.ldwq; p.q_colsum = a.q_colsum; p.q_bias = a.q_bias;
  p.kvpack = a.kvpack; p.rows_per_sample = a.rows_per_sample;
  p.wo = a.wo; p.ldwo = a.ldwo; p.o_bias = a.o_bias;
  p.out = a.out; p.ldo = a.ldo; p.ln_stats_out = a.ln_stats_out; p.Nk = a.Nk;
  // algorithmic work of the three launches it replaces: two [M, 320] x [320, 320] projections + the attention
  const double flops = 2.0 * 2.0 * a.M * (double)xf::C * xf::C + 4.0 * a.M * (double)a.Nk * xf::C;
  AfProfScope prof(AF_K_ATTENTION, stream, flops, 2.0 * a.M * (double)xf::C * 2);
  hipLaunchKernelGGL(xf::xattn_fused_kernel, dim3(a.M / xf::BM), dim3(512), xf::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  g_af_xattn_fused_launches += 1;
  return 0;
}
