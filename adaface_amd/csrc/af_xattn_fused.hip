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
//   phase 2  per head: S^T = K q^T    K PACKED per layer at af_set_context (as stored; scale * log2(e) enters in the fma in front of exp2) in exactly that
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
// phase 3 (to_out): 160-column tiles through a three-slot ring (prefetch distance 2), as rowpanel_kernel's plain form
constexpr int BN3 = 160, NIW3 = 10, NTILE3 = 2, D3 = 2, NS3 = 3, WBYTES3 = BN3 * 128, WPIECES3 = BN3 / 8;
constexpr int OPITCH = BN3 * 2 + 16;                                        // per-wave transposition tile: 32 rows x 336 B ...
constexpr int OCH = BN3 / 8, OSLOTS = OCH + 1;                              // 20 data chunks + 1 pad chunk per row
constexpr int NST = WROWS * OCH / 64;                                       // 10 row-segment stores per lane and tile
constexpr int NRES = (WROWS * OSLOTS + 63) / 64;                            // 11 LDS-DMA pieces bring a tile's residual rows
constexpr int OBYTES = NRES * 1024;                                         // ... rounded up to whole pieces (no lane of a piece is masked)
constexpr int RING_BYTES = NS3 * WBYTES3;                                   // 60 KiB (phase 1 uses 5 x 10 KiB of it)
constexpr int RING0 = 0, KV0 = RING_BYTES, KV_REGION = 8 * OBYTES;          // K / V buffers, later the transposition tiles
constexpr int VEC0 = KV0 + KV_REGION;
constexpr int LDS_BYTES = VEC0 + 3 * C * 4;          // + the column sums / folded bias of to_q and the bias of to_out (fp32)
static_assert(NS * WBYTES <= RING_BYTES && 2 * KVBUF <= KV_REGION && WROWS * OPITCH <= OBYTES && (WROWS * OCH) % 64 == 0 &&
              LDS_BYTES <= 160 * 1024, "LDS plan");

// packed K / V of one (sample, head pair): KV_PPW * 8 fragments of [64 lanes][8 bf16]
//   K fragment (hh, kb, s)  at ((hh * NKB + kb) * 2 + s):   lane (key = 16 kb + (lane & 15), g = lane >> 4), element j
//   V fragment (hh, blk, sp) at K_FRAGS + (hh * 3 + blk) * NSP + sp: lane (row = lane & 15 of block blk, g), element j
constexpr long PACK_ELEMS_PER_PAIR = (long)KV_PPW * 8 * 512;

// Operand tuples.  q (phase 1) and then O (phase 2, in place) live as MFMA B operands of four VGPRs = two 16-channel
// accumulator blocks of one token row block: per head pair p (columns 80 p .. 80 p + 79 = blocks t = 0..4 of 16)
//     qA[p] = (t0, t1) = first head, d 0-31        qB[p] = (t3, t4) = second head, d 8-39
//     qC[p / 2] = (t2 of pair 2 (p / 2), t2 of pair 2 (p / 2) + 1): first head d 32-39 (rows 0-7) | second head d 0-7 (rows 8-15)
// so that no operand is ever assembled from two places (v_mov pairs per MFMA otherwise).  A lane holds rows 4 g + e of a block.
//
// channel d (0..39) of head hh (0 / 1 of the pair) that K slot (s, g, j) of the S^T contraction multiplies, or -1 = zero:
// k step 0 = the head's own tuple, k step 1 = the shared tuple, of which pair p owns half (p & 1)
__host__ __device__ inline int k_slot_channel(int pair, int hh, int s, int g, int j) {
  const int r = 4 * g + (j & 3);
  if (s == 0) return (hh == 0 ? 0 : 8) + 16 * (j >> 2) + r;
  if ((j >> 2) != (pair & 1)) return -1;             // the other pair's half of the shared tuple
  return hh == 0 ? (r < 8 ? 32 + r : -1) : (r >= 8 ? r - 8 : -1);
}
// channel d held by row r of O^T block blk (0, 1: the head's own tuple; 2: its rows of the shared block) of head hh, or -1
__host__ __device__ inline int v_row_channel(int hh, int blk, int r) {
  if (blk < 2) return (hh == 0 ? 0 : 8) + 16 * blk + r;
  return hh == 0 ? (r < 8 ? 32 + r : -1) : (r >= 8 ? r - 8 : -1);
}
// key held by K slot (g, j) of PV k step sp
__host__ __device__ inline int pv_slot_key(int sp, int g, int j) { return 32 * sp + 16 * (j >> 2) + 4 * g + (j & 3); }
// input channel (0..319) of to_out at position pos (0..63) of K chunk kc of the permuted weight.  Phase 3 contracts over the
// ten tuples in the order qA0, qB0, qA1, qB1, qA2, qB2, qA3, qB3, qC0, qC1 (two per chunk)
__host__ __device__ inline int wo_pos_channel(int kc, int pos) {
  const int u = pos >> 5, g = (pos >> 3) & 3, j = pos & 7;
  const int o = 2 * kc + u, half = j >> 2, r = 4 * g + (j & 3);
  if (o < 8) {
    const int pair = o >> 1, hh = o & 1;
    return 80 * pair + 40 * hh + (hh == 0 ? 0 : 8) + 16 * half + r;
  }
  const int pair = 2 * (o - 8) + half;
  return r < 8 ? 80 * pair + 32 + r : 80 * pair + 40 + (r - 8);
}

__global__ __launch_bounds__(256) void pack_kv_kernel(const bf16* __restrict__ kv, int ldk, long bsk, int Nk, int B,
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
      const int key = 16 * kb + l15, d = k_slot_channel(pair, hh, s, g, j);
      if (key < Nk && d >= 0) val = (float)kv[(long)b * bsk + (long)key * ldk + (2 * pair + hh) * DH + d];   // (as stored: scale * log2 e enters in front of the exponential)
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
__device__ __forceinline__ unsigned pack2(float a, float b) {   // one v_cvt_pk_bf16_f32 (element 0 = low half)
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{a, b}, bf16x2));
}
// one v_max3_f32 (plain fmaxf on MFMA results makes hipcc add a canonicalising v_max per operand).  NOPS > 0: wait states in
// front of it -- the asm reads MFMA results, and the hazard recogniser does not look inside inline asm (4-pass MFMA -> VALU
// read; the S^T MFMAs of the block were issued at least two MFMAs earlier).  volatile: the others stay behind that one
template <int NOPS> __device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  if constexpr (NOPS > 0) asm volatile("s_nop %4\n\tv_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c), "n"(NOPS - 1));
  else asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
// the two bf16 halves of a dword as f32 (element 0 = low half)
__device__ __forceinline__ f32x2 unpack2(unsigned w) {
  return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xFFFF0000u)};
}
// max over the four lanes l, l ^ 16, l ^ 32, l ^ 48 (the lane groups that hold one token).  Two ds_bpermute round trips: a first
// version did it on the VALU (v_permlane16_swap + v_permlane32_swap with the two wait states LLVM's table asks for in front) and was
// not reproducible -- 5 forwards in 300 differed by an ulp of P somewhere; with the shuffles 0 in 480 (scripts/lab/dbg_det.py).
// Per head and row block it is two LDS operations against ~75 vector instructions: not measurable (56.8 vs 57.2 us per launch).
__device__ __forceinline__ float xgroup_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
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
  float sl2;                             // dh^-1/2 * log2(e): multiplies the raw scores in the fma in front of exp2
  int lab;                               // -DAF_LAB_ABLATE builds only: timing ablations (wrong results), see scripts/lab/ablate_xattn.sh
};

__global__ __launch_bounds__(512) void xattn_fused_kernel(const Params p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AF_LAB_ABLATE
  const int lab = p.lab;   // 1: no phase-1 steps, 2: no attention, 4: no phase-3 steps, 8: no row loads, 16: no epilogue stores / residual
#else
  constexpr int lab = 0;
#endif
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
        xr[j][kc][u] = (lab & 8) ? u32x4{0u, 0u, 0u, 0u} : __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_x, xo, (kc * 64 + u * 32) * 2, 0));
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

  // ---- weight ring of phase 1: step t = (80-column tile t / KC, chunk t % KC) of to_q ----
  const int srow = lane >> 3;
  const unsigned dchunk = (unsigned)((lane & 7) ^ srow);
  const int nwq = wid < WPIECES - 8 ? 2 : 1;         // pieces per wave and step (10 pieces over 8 waves)
  unsigned wq_off[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) wq_off[q] = (unsigned)(((wid + 8 * q) * 8 + srow) * p.ldwq * 2) + dchunk * 16u;
  const unsigned tstride_q = (unsigned)(BNT * p.ldwq * 2);
  constexpr int T1 = NTILE * KC;                     // phase-1 steps; a step past the end stages zeros (keeps the vmcnt arithmetic uniform)
  auto stage = [&](int t, int slot) {
    const bool live = t < T1;
    const int ntl = t / KC, kc = t - ntl * KC;
    const unsigned so = (unsigned)ntl * tstride_q + (unsigned)kc * 128u;
    lds_dma16(rs_wq, smem + RING0 + slot * WBYTES + wid * 1024, live ? wq_off[0] : 0xFFFFFFFFu, so);
    if (wid < WPIECES - 8) lds_dma16(rs_wq, smem + RING0 + slot * WBYTES + (wid + 8) * 1024, live ? wq_off[1] : 0xFFFFFFFFu, so);
  };
  // phase 3: to_out in 160-column tiles, 20 pieces per step (waves 0-3: three, waves 4-7: two)
  unsigned wo_off[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) wo_off[q] = (unsigned)(((wid + 8 * q) * 8 + srow) * p.ldwo * 2) + dchunk * 16u;
  const unsigned tstride_o = (unsigned)(BN3 * p.ldwo * 2);
  const int nw3 = wid < WPIECES3 - 16 ? 3 : 2;
  constexpr int T3 = NTILE3 * KC;
  auto stage3 = [&](int t, int slot) {
    const bool live = t < T3;
    const int ntl = t / KC, kc = t - ntl * KC;
    const unsigned so = (unsigned)ntl * tstride_o + (unsigned)kc * 128u;
    lds_dma16(rs_wo, smem + RING0 + slot * WBYTES3 + wid * 1024, live ? wo_off[0] : 0xFFFFFFFFu, so);
    lds_dma16(rs_wo, smem + RING0 + slot * WBYTES3 + (wid + 8) * 1024, live ? wo_off[1] : 0xFFFFFFFFu, so);
    if (wid < WPIECES3 - 16) lds_dma16(rs_wo, smem + RING0 + slot * WBYTES3 + (wid + 16) * 1024, live ? wo_off[2] : 0xFFFFFFFFu, so);
  };
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned fch0 = (unsigned)(g ^ (lane & 7)) * 16u;
  const unsigned fch1 = (unsigned)((g + 4) ^ (lane & 7)) * 16u;
  const unsigned w_base = (unsigned)(l15 * 128);
  const int cl = 4 * g;

  f32x4 acc[NIW][MJ];
  u32x4 wf[NIW][2];
  // one K step of a projection tile: the B operand of row block j is (b0[j], b1[j]); weight blocks read three ahead
  auto step = [&](auto firstc, int slot, int t, int stage_slot, const u32x4 (&b0)[MJ], const u32x4 (&b1)[MJ]) {
    constexpr bool first = decltype(firstc)::value;   // first K step of a tile: C = 0 in the instruction, no zeroing of 40 registers
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
        acc[i][j] = mma(wf[i][0], b0[j], first ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j]);
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
  auto step_head = [&]() {
    // own pieces of this step landed: everything but the pieces of the next D - 1 steps
    if (nwq == 2) wait_vm<2 * (D - 1)>(); else wait_vm<(D - 1)>();
    __builtin_amdgcn_s_barrier();
  };

  // =========================== phase 1: q (bf16) as operand tuples (see k_slot_channel) ===========================
  u32x4 qA[NPAIR][MJ], qB[NPAIR][MJ], qC[NPAIR / 2][MJ];
  static_for<0, NTILE>([&](auto ntc) {
    constexpr int nt = decltype(ntc)::value;
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      if (lab & 1) return;
      step_head();
      const int prev = cur == 0 ? NS - 1 : cur - 1;
      const u32x4 b0[MJ] = {xr[0][kc][0], xr[1][kc][0]}, b1[MJ] = {xr[0][kc][1], xr[1][kc][1]};
      step(std::integral_constant<bool, kc == 0>{}, cur, nt * KC + kc, prev, b0, b1);
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
      });
      u32x2 qb[NIW][MJ];
      static_for<0, NIW>([&](auto ic) {
        constexpr int k = decltype(ic)::value;
        wait_lgkm2<2 * (NIW - 1 - k)>(csv[k], bvv[k]);
        const f32x4 cs = __builtin_bit_cast(f32x4, csv[k]), bv = __builtin_bit_cast(f32x4, bvv[k]);
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
          f32x4 v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = (acc[k][j][e] - ln_mu[j] * cs[e]) * ln_rs[j] + bv[e];
          qb[k][j] = pack4(v);
        }
      });
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        qA[nt][j] = cat(qb[0][j], qb[1][j]);
        qB[nt][j] = cat(qb[3][j], qb[4][j]);
        if constexpr ((nt & 1) == 0) { qC[nt / 2][j].x = qb[2][j].x; qC[nt / 2][j].y = qb[2][j].y; qC[nt / 2][j].z = 0u; qC[nt / 2][j].w = 0u; }
        else { qC[nt / 2][j].z = qb[2][j].x; qC[nt / 2][j].w = qb[2][j].y; }
        // (defined HERE: hipcc otherwise sinks the fold + conversion to the use in phase 2 and keeps the fp32 accumulators alive)
        asm volatile("" : "+v"(qA[nt][j]), "+v"(qB[nt][j]), "+v"(qC[nt / 2][j]));
      }
    }
  });

  // =========================== phase 2: attention, head pair by head pair; O replaces q in its tuples ===========================
  // (every wave is past its last read of the phase-1 ring once it passes this barrier, and -- vmcnt(0) -- none of its own
  // pieces is still on its way to a slot: to_out's first two steps may land there.  Barriers do not drain LDS-DMA)
  wait_vm<0>();
  __builtin_amdgcn_s_barrier();
  if (!(lab & 4)) { stage3(0, 0); stage3(1, 1); }
  u32x4 ones;
  ones = u32x4{0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u};
  asm volatile("" : "+v"(ones));
  // keys >= Nk (64 < Nk <= 80: all in the last key block): an additive (0 / -inf) vector for this lane's four rows of it
  f32x4 pmask;
#pragma unroll
  for (int e = 0; e < 4; ++e) pmask[e] = (16 * (NKB - 1) + 4 * g + e >= p.Nk) ? -INFINITY : 0.f;
  static_for<0, NPAIR>([&](auto pc) {
    constexpr int pr = decltype(pc)::value;
    if (lab & 2) return;
    if constexpr (pr == 2) { wait_vm<KV_PPW>(); __builtin_amdgcn_s_barrier(); }   // own pieces of pair 2 landed (pair 3's may fly)
    if constexpr (pr == 3) { wait_vm<0>(); __builtin_amdgcn_s_barrier(); }
    const unsigned kvb = lds0 + (unsigned)(KV0 + (pr & 1) * KVBUF) + (unsigned)lane * 16u;
    f32x4 oc[MJ];                                   // the O^T block the two heads share (rows 0-7: head 0, rows 8-15: head 1)
    float inv0[MJ];
    static_for<0, 2>([&](auto hc) {
      constexpr int hh = decltype(hc)::value;
      // ---- S^T = K q^T: NKB key blocks x 2 k steps (own tuple, shared tuple), both row blocks of the wave ----
      f32x4 sc[MJ][NKB];
      {
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
            sc[j][kb] = mma(kf[kb][0], hh == 0 ? qA[pr][j] : qB[pr][j], f32x4{0.f, 0.f, 0.f, 0.f});
            sc[j][kb] = mma(kf[kb][1], qC[pr / 2][j], sc[j][kb]);
          }
        });
        // every S^T MFMA is ISSUED here: hipcc otherwise sinks some of them into the v_max3 chains below, which read their
        // results from inline asm -- no hazard padding there -- a few cycles too early on the younger wave of a SIMD (seen: the
        // row maximum, and with it the bf16 rounding of P, changed from run to run on waves 4-7)
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
          for (int kb = 0; kb < NKB; ++kb) asm volatile("" : "+v"(sc[j][kb]));
      }
      // V^T fragments of this head (own blocks 0, 1 + its rows of the shared block): read once, used by both row blocks
      constexpr unsigned vbase = (unsigned)((K_FRAGS + hh * 3 * NSP) * 1024);
      u32x4 vf[3][NSP];
      static_for<0, 3 * NSP>([&](auto fc) {
        constexpr int f = decltype(fc)::value;
        vf[f / NSP][f % NSP] = lds_read128<vbase + f * 1024>(kvb);
      });
      // ---- row block by row block: the exact softmax (VALU) of block j runs under the MFMAs issued just before it -- the S^T
      // of block 1 / the PV of block 0 -- and both blocks are normalised after the second PV ----
      f32x4 o0[MJ], o1[MJ], ls[MJ];
      static_for<0, MJ>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        sc[j][NKB - 1] += pmask;
        // the row maximum: v_max3 chains straight on the MFMA results (the first one carries the wait states)
        float mx;
        {
          static_assert(NKB == 5, "max tree written for 20 scores per lane");
          const float m0 = max3f<8>(sc[j][0][0], sc[j][0][1], sc[j][0][2]);
          const float m1 = max3f<0>(sc[j][0][3], sc[j][1][0], sc[j][1][1]);
          const float m2 = max3f<0>(sc[j][1][2], sc[j][1][3], sc[j][2][0]);
          const float m3 = max3f<0>(sc[j][2][1], sc[j][2][2], sc[j][2][3]);
          const float m4 = max3f<0>(sc[j][3][0], sc[j][3][1], sc[j][3][2]);
          const float m5 = max3f<0>(sc[j][3][3], sc[j][4][0], sc[j][4][1]);
          const float m6 = max3f<0>(sc[j][4][2], sc[j][4][3], m0);
          const float m7 = max3f<0>(m1, m2, m3);
          mx = xgroup_max(max3f<0>(m4, m5, max3f<0>(m6, m7, m7)));
        }
        u32x4 pb[NSP];
        {
          const float msl = mx * p.sl2;
          const f32x2 sl2v = f32x2{p.sl2, p.sl2}, nm2 = f32x2{-msl, -msl};
#pragma unroll
          for (int sp = 0; sp < NSP; ++sp) {
            unsigned w[4];
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
              const int kb = 2 * sp + hb;
              if (kb < NKB) {
                const f32x2 d0 = __builtin_elementwise_fma(f32x2{sc[j][kb][0], sc[j][kb][1]}, sl2v, nm2);   // v_pk_fma_f32
                const f32x2 d1 = __builtin_elementwise_fma(f32x2{sc[j][kb][2], sc[j][kb][3]}, sl2v, nm2);
                w[2 * hb] = pack2(__builtin_amdgcn_exp2f(d0.x), __builtin_amdgcn_exp2f(d0.y));
                w[2 * hb + 1] = pack2(__builtin_amdgcn_exp2f(d1.x), __builtin_amdgcn_exp2f(d1.y));
              } else {
                w[2 * hb] = w[2 * hb + 1] = 0u;
              }
            }
            pb[sp] = u32x4{w[0], w[1], w[2], w[3]};
          }
        }
        // O^T = V^T P^T and the denominator (ones x P^T: sums the bf16-rounded P the numerator uses)
        if constexpr (j == 0) {          // every V^T fragment is back (issued before the softmax); the wait names all of them
          static_assert(NSP == 3, "nine V^T fragments");
          asm volatile("s_waitcnt lgkmcnt(0)"
                       : "+v"(vf[0][0]), "+v"(vf[0][1]), "+v"(vf[0][2]), "+v"(vf[1][0]), "+v"(vf[1][1]), "+v"(vf[1][2]), "+v"(vf[2][0]),
                         "+v"(vf[2][1]), "+v"(vf[2][2])
                       :
                       : "memory");
        }
        static_for<0, NSP>([&](auto spc) {
          constexpr int sp = decltype(spc)::value;
          ls[j] = mma(ones, pb[sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : ls[j]);
          o0[j] = mma(vf[0][sp], pb[sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o0[j]);
          o1[j] = mma(vf[1][sp], pb[sp], sp == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : o1[j]);
          oc[j] = mma(vf[2][sp], pb[sp], (hh == 0 && sp == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : oc[j]);
        });
      });
      // normalise; O tuples take the place of the q tuples they came from
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const float inv = __builtin_amdgcn_rcpf(ls[j][0]);
        const u32x4 ot = cat(pack4(o0[j] * inv), pack4(o1[j] * inv));
        if constexpr (hh == 0) {
          qA[pr][j] = ot;
          asm volatile("" : "+v"(qA[pr][j]));
          inv0[j] = inv;
        } else {
          qB[pr][j] = ot;
          const u32x2 sh = pack4(oc[j] * (g < 2 ? inv0[j] : inv));
          if constexpr ((pr & 1) == 0) { qC[pr / 2][j].x = sh.x; qC[pr / 2][j].y = sh.y; }
          else { qC[pr / 2][j].z = sh.x; qC[pr / 2][j].w = sh.y; }
          asm volatile("" : "+v"(qB[pr][j]), "+v"(qC[pr / 2][j]));
        }
      }
    });
    if constexpr (pr < 2) {                          // refill this pair's buffer with pair + 2 once every wave has left it
      __builtin_amdgcn_s_barrier();
      stage_kv(pr + 2);
    }
  });

  // =========================== phase 3: out = to_out(O) + bias + x ===========================
  // 160-column tiles (ten 16-column blocks), three 20-KiB ring slots.  The wave's transposition tile (32 rows x 336 B, where the
  // K / V buffers were) first RECEIVES the residual rows of the tile by LDS-DMA -- whole 16-byte chunks of whole rows, issued a
  // tile ahead -- then takes the bf16 results in the accumulator layout in their place and leaves as 16-byte row stores.
  char* otile = smem + KV0 + wid * OBYTES;
  const unsigned ldx2 = (unsigned)p.ldx * 2u, ldo2 = (unsigned)p.ldo * 2u;
  auto stage_res = [&](int nt) {   // NRES whole pieces: slot idx = 64 i + lane -> (row idx / 21, chunk idx % 21); the pad chunk of a
#pragma unroll                     // row and the slots behind the last row fetch out of range (zeros)
    for (int i = 0; i < NRES; ++i) {
      const unsigned idx = (unsigned)(i * 64 + lane), row = idx / OSLOTS, c = idx - row * OSLOTS;
      const unsigned off = (row < WROWS && c < OCH) ? (unsigned)(r0 + (int)row) * ldx2 + c * 16u : 0xFFFFFFFFu;
      lds_dma16(rs_x, otile + i * 1024, off, (unsigned)(nt * BN3 * 2));
    }
  };
  // the accumulators START from the bias of their four columns (20 LDS reads per tile instead of zeroing 80 registers and
  // adding the bias in the epilogue)
  f32x4 acc3[NIW3][MJ];
  auto init_acc3 = [&](int ncol) {
    const unsigned va = lds0 + (unsigned)(VEC0 + 2 * C * 4) + 16u * (unsigned)g + (unsigned)ncol * 4u;
    static_for<0, NIW3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
#pragma unroll
      for (int j = 0; j < MJ; ++j) acc3[i][j] = __builtin_bit_cast(f32x4, lds_read128<i * 64>(va));
    });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NIW3; ++i)
#pragma unroll
      for (int j = 0; j < MJ; ++j) asm volatile("" : "+v"(acc3[i][j]));
  };
  init_acc3(0);
  u32x4 wf3[NIW3][2];
  auto step3 = [&](int slot, int t, int stage_slot, const u32x4 (&b0)[MJ], const u32x4 (&b1)[MJ]) {
    const unsigned a0 = lds0 + (unsigned)(RING0 + slot * WBYTES3) + w_base + fch0;
    const unsigned a1 = lds0 + (unsigned)(RING0 + slot * WBYTES3) + w_base + fch1;
    auto rd = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      wf3[i][0] = lds_read128<i * 2048>(a0);
      wf3[i][1] = lds_read128<i * 2048>(a1);
    };
    rd(std::integral_constant<int, 0>{});
    rd(std::integral_constant<int, 1>{});
    rd(std::integral_constant<int, 2>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<0, NIW3>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      constexpr int issued = (i + 3) < NIW3 ? (i + 3) : NIW3;
      wait_lgkm2<2 * (issued - i - 1)>(wf3[i][0], wf3[i][1]);
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        acc3[i][j] = mma(wf3[i][0], b0[j], acc3[i][j]);
        acc3[i][j] = mma(wf3[i][1], b1[j], acc3[i][j]);
        asm volatile("" : "+v"(acc3[i][j]));
      }
      if constexpr (i + 3 < NIW3) rd(std::integral_constant<int, i + 3>{});
      if constexpr (i == 1) stage3(t + D3, stage_slot);
      __builtin_amdgcn_sched_barrier(0);
    });
    wait_lgkm0();
  };
  // vmcnt at the head of a phase-3 step: everything but the pieces of the next step -- and, while they are younger than the
  // pieces waited for, the previous epilogue's NST stores + the NRES residual pieces issued behind them (tile 0: the residual
  // pieces only, issued at the head of its first step)
  auto step_head3 = [&](int extra) {
    if (extra == 2) { if (nw3 == 3) wait_vm<3 + NST + NRES>(); else wait_vm<2 + NST + NRES>(); }
    else if (extra == 1) { if (nw3 == 3) wait_vm<3 + NRES>(); else wait_vm<2 + NRES>(); }
    else { if (nw3 == 3) wait_vm<3>(); else wait_vm<2>(); }
    __builtin_amdgcn_s_barrier();
  };
  int cur3 = 0;
  static_for<0, NTILE3>([&](auto ntc) {
    constexpr int nt = decltype(ntc)::value;
    static_for<0, KC>([&](auto kcc) {
      constexpr int kc = decltype(kcc)::value;
      if (lab & 4) return;
      // (tile 0, step 0: the residual pieces go out behind the barrier -- every wave has left the K / V buffers; step 1 then
      // still sees them in front of the pieces it waits for.  Later tiles: issued by the epilogue, D3 heads see stores + pieces)
      step_head3(nt == 0 ? (kc == 1 ? 1 : 0) : (kc < D3 ? 2 : 0));
      if constexpr (nt == 0 && kc == 0) { if (!(lab & 16)) stage_res(0); }
      const int prev = cur3 == 0 ? NS3 - 1 : cur3 - 1;
      // K chunk kc = tuples 2 kc, 2 kc + 1 of (qA0, qB0, qA1, qB1, qA2, qB2, qA3, qB3, qC0, qC1): wo_pos_channel
      if constexpr (kc < 4) {
        const u32x4 b0[MJ] = {qA[kc][0], qA[kc][1]}, b1[MJ] = {qB[kc][0], qB[kc][1]};
        step3(cur3, nt * KC + kc, prev, b0, b1);
      } else {
        const u32x4 b0[MJ] = {qC[0][0], qC[0][1]}, b1[MJ] = {qC[1][0], qC[1][1]};
        step3(cur3, nt * KC + kc, prev, b0, b1);
      }
      cur3 = cur3 + 1 == NS3 ? 0 : cur3 + 1;
    });
    // ---- epilogue of the 160-column tile: + residual (from the tile; the bias is in the accumulators), bf16 in place,
    // LayerNorm partial sums, row stores.  (The residual pieces are older than the last step's waited-for weight pieces: landed.)
    constexpr int ncol = nt * BN3;
    if (lab & 16) return;
    // (every row of the tile exists: M is a multiple of 256.)  Packed f32 arithmetic on pairs: residual pair from the tile
    // (two bf16 = one dword) + accumulator, one v_cvt_pk, the rounded pair back for the LayerNorm sums
    float ps[MJ], pq[MJ];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
      f32x2 ps2 = f32x2{0.f, 0.f}, pq2 = f32x2{0.f, 0.f};
      char* trow = otile + (j * 16 + l15) * OPITCH + cl * 2;
      static_for<0, 2>([&](auto hc2) {
        constexpr int hf = decltype(hc2)::value;      // five blocks at a time: their ten residual dwords in flight together
        u32x2 rq[NIW3 / 2];
#pragma unroll
        for (int ii = 0; ii < NIW3 / 2; ++ii) rq[ii] = *reinterpret_cast<const u32x2*>(trow + (hf * (NIW3 / 2) + ii) * 32);
#pragma unroll
        for (int ii = 0; ii < NIW3 / 2; ++ii) {
          const int i = hf * (NIW3 / 2) + ii;
          const f32x2 v01 = f32x2{acc3[i][j][0], acc3[i][j][1]} + unpack2(rq[ii].x);
          const f32x2 v23 = f32x2{acc3[i][j][2], acc3[i][j][3]} + unpack2(rq[ii].y);
          const unsigned w01 = pack2(v01.x, v01.y), w23 = pack2(v23.x, v23.y);
          const f32x2 r01 = unpack2(w01), r23 = unpack2(w23);       // the values the consumer will read
          ps2 += r01; ps2 += r23;
          pq2 += r01 * r01; pq2 += r23 * r23;
          *reinterpret_cast<u32x2*>(trow + i * 32) = u32x2{w01, w23};
        }
      });
      ps[j] = ps2.x + ps2.y;
      pq[j] = pq2.x + pq2.y;
    }
    // per-row partial sums in the layout of the unfused to_out launch (row-panel, 160-column tiles): parts 2 nt hold the sums of
    // this tile's columns, parts 2 nt + 1 are zero
    if (p.ln_stats_out) {
#pragma unroll
      for (int j = 0; j < MJ; ++j) {
        const int m = r0 + j * 16 + l15;
        float s1 = ps[j], s2 = pq[j];
        s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (g == 0) {
          *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(2 * nt) * p.M + m) * 2) = float2{s1, s2};
          *reinterpret_cast<float2*>(p.ln_stats_out + ((long)(2 * nt + 1) * p.M + m) * 2) = float2{0.f, 0.f};
        }
      }
    }
    // (wave-private tile: no barrier; all chunks first, then the stores back to back: a 16-byte buffer store with an SGPR offset
    // must not be followed by a VALU write of its data registers, scripts/check_isa_hazards.py)
    {
      u32x4 chunk[NST];
      unsigned orow[NST];
#pragma unroll
      for (int i = 0; i < NST; ++i) {
        const unsigned c = (unsigned)(lane + 64 * i), row = c / OCH, ch = c - row * OCH;
        orow[i] = (unsigned)(r0 + (int)row) * ldo2 + ch * 16u;
        chunk[i] = *reinterpret_cast<const u32x4*>(otile + row * OPITCH + ch * 16);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NST; ++i) __builtin_amdgcn_raw_buffer_store_b128(chunk[i], rs_o, orow[i], ncol * 2, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (nt + 1 < NTILE3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the tile's chunks are in registers before the next residual lands on it)
      stage_res(nt + 1);
      init_acc3((nt + 1) * BN3);
    }
  });
  // every LDS-DMA of this wave has landed (the last tile's NST stores are the only younger operations; they may still fly)
  wait_vm<NST>();
}
}  // namespace xf

// ---------------------------------------------------------------------------------------------------------------------------
long af_xattn_fused_pack_elems(int B, int H, int dh, int Nk) {
  if (H != xf::H || dh != xf::DH || Nk <= xf::SMAX - 16 || Nk > xf::SMAX || B <= 0) return 0;
  return (long)B * xf::NPAIR * xf::PACK_ELEMS_PER_PAIR;
}
int af_launch_xattn_fused_pack(const void* kv, int ldk, long bsk, int Nk, int B, void* pack, hipStream_t stream) {
  const long total = (long)B * xf::NPAIR * xf::PACK_ELEMS_PER_PAIR;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  // (round 4: K is packed as stored -- no second bf16 rounding; the kernel multiplies the scores by dh^-1/2 * log2 e in fp32)
  hipLaunchKernelGGL(xf::pack_kv_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16*>(kv), ldk, bsk, Nk, B,
                     reinterpret_cast<bf16*>(pack));
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
  // (keys: all masking happens in the last key block of 16, so 64 < Nk <= 80 -- the text encoder's 77)
  return C == xf::C && H == xf::H && dh == xf::DH && Nk > xf::SMAX - 16 && Nk <= xf::SMAX && rows_per_sample % xf::BM == 0 && M % xf::BM == 0 && M > 0;
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
  p.sl2 = 1.0f / sqrtf((float)xf::DH) * 1.44269504088896340736f;   // (the expression of the unfused kernels: p.scale * log2 e)
  p.lab = g_af_knobs.xattn_fused >> 4;
  // algorithmic work of the three launches it replaces: two [M, 320] x [320, 320] projections + the attention
  const double flops = 2.0 * 2.0 * a.M * (double)xf::C * xf::C + 4.0 * a.M * (double)a.Nk * xf::C;
  AfProfScope prof(AF_K_ATTENTION, stream, flops, 2.0 * a.M * (double)xf::C * 2);
  hipLaunchKernelGGL(xf::xattn_fused_kernel, dim3(a.M / xf::BM), dim3(512), xf::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  g_af_xattn_fused_launches += 1;
  return 0;
}
