// Flash-style attention for gfx950: softmax(Q K^T * scale) V without materialising
// the [N, S] score matrix.  Serves both uses of the reference's CrossAttention
// class (ldm/modules/attention.py:172-257): attn1 = self-attention over
// N in {4096,1024,256,64} image tokens and attn2 = cross-attention over the 77
// context tokens; 8 heads, dh = C/8 in {40,80,160}.  The reference computes
// einsum -> softmax -> einsum with a full `sim` tensor (attention.py:199,238,240).
//
// Structure (wave64, MFMA 32x32):
//   * grid (ceil(Nq/128), heads, batch); 4 waves, each owns 32 query rows.
//   * K/V staged per 64-key tile in LDS.  bf16: both row-major, double-buffered, the
//     next tile's global loads are issued into registers before the current tile's
//     MFMAs and written to the other buffer after them (one barrier per tile).
//     K rows are padded by 16 B (conflict-free ds_read_b128); V rows use a stride
//     = 64 or 192 (mod 256) so the hardware-transposing ds_read_b64_tr_b16 that
//     feeds V^T as the MFMA A operand is conflict-free.  f32 (parity mode): single
//     buffer, V staged transposed (no 32-bit transposing read exists).
//   * swapped QK^T: S^T = K . Q^T, so a lane holds ONE query column and 16 keys
//     per 32x32 block in registers; the online-softmax max/sum are register
//     reductions plus one v_permlane32_swap across the two lane halves.
//   * Q is pre-multiplied by scale*log2(e) once, so softmax is exp2(s - m) with a
//     bare v_exp_f32; keys are masked only in the last (partial) tile; the O
//     accumulator is rescaled only when some lane's running max actually moved.
//   * the S^T accumulator is fed straight back as the B operand of O^T = V^T . P^T
//     (accumulator-as-operand: register r of lane-half h is key (r&3)+8(r>>2)+4h,
//     the V^T fragment is read in that same key order), no LDS round trip for P.
//   * dh is padded in LDS only: QK^T K-dim to a multiple of 32 bytes, O to
//     32-row blocks; HBM traffic is exactly the unpadded Q, K, V, O.
#include "af_common.h"
#include <math.h>
#include <type_traits>

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

template <typename T, int DH> struct AttnCfg {
  static constexpr bool BF = sizeof(T) == 2;
  static constexpr int EPC = 16 / sizeof(T);
  static constexpr int FS = (DH * (int)sizeof(T) + 31) / 32;   // 32-byte steps along d
  static constexpr int KROW = FS * 32 + 16;                    // bytes
  static constexpr int DB = (DH + 31) / 32;                    // 32-wide d blocks
  // bf16: V row-major [64 keys][VROW], VROW >= DB*64 and == 64 or 192 (mod 256)
  static constexpr int VROW_BF = (DB * 64 <= 64) ? 64 : (DB * 64 <= 192) ? 192 : 320;
  // f32: V transposed [DB*32][64 keys * 4 + 16]
  static constexpr int VROW_F32 = 64 * 4 + 16;
  static constexpr int K_BYTES = 64 * KROW;
  static constexpr int V_BYTES = BF ? 64 * VROW_BF : DB * 32 * VROW_F32;
  static constexpr int TILE = K_BYTES + V_BYTES;
  static constexpr int NBUF = BF ? 2 : 1;
  static constexpr int LDS_BYTES = NBUF * TILE;
  static constexpr int CPR = DH / EPC;                         // valid 16-byte chunks per key row
  static constexpr int NLD = (64 * CPR + 255) / 256;           // staged chunks per thread (K and V each)
  // a free padding column d = DH of the V tile holds 1.0 for valid keys: row DH of O^T is then the softmax
  // denominator, summed by the MFMA instead of 32 VALU adds per tile (bf16 only; needs DH % 32 != 0)
  static constexpr bool ONES = BF && (DH % 32) != 0;
  // a free padding slot d = DH in the QK^T K-dimension (dh = 40: 80 -> 96 bytes) carries "1.0" on the K side and
  // "-m_ref" (the per-query running softmax reference, kept bf16-representable) on the Q side: the MFMA then returns
  // s - m_ref directly and the 32 v_sub per tile disappear; m_ref moves only on the first tile or when a score rises
  // 2^24 above it.  With ONE exp loop behind a rare "move the reference" fix-up the kernel needs 132 VGPRs (the first
  // version with two unrolled loops needed 190 and lost); dh = 40 self-attention 612 -> 539 us, 515 us at 4 waves/SIMD.
  static constexpr bool MREF = BF && ((DH * 2) % 32) != 0;
  static constexpr int MREF_STEP = (DH * 2) / 32, MREF_HALF = ((DH * 2) % 32) / 16, MREF_ELEM = (((DH * 2) % 32) % 16) / 2;
};

// max over the two lane halves (lane l <-> lane l ^ 32).  v_permlane32_swap exchanges the upper half of its first operand
// with the lower half of its second: afterwards a = (low | low) and b = (high | high) of the input.  The swap is inline
// asm on purpose: through __builtin_amdgcn_permlane32_swap hipcc (ROCm 7.2) treats the two results as one value and folds
// fmaxf(r[0], r[1]) to r[0] — every lane then got the LOWER half's maximum only, which went unnoticed on ordinary data
// (any reference below the true maximum is still a valid softmax shift) and produced inf / NaN once a key held by the upper
// half scored 2^128 above the rest (tests/test_ops_gpu.py::test_attention_spiky_bf16_dh40).  The s_nop 1 covers the
// "VALU write -> v_permlane read" hazard for whatever wrote the operands.
__device__ __forceinline__ float xhalf_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}

// smallest bf16-representable value >= x, as fp32
__device__ __forceinline__ float bf16_ceil(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  if (x > 0.f && (u & 0xFFFFu)) u += 0x10000u;  // positive: bump; negative: truncation already moves up
  u &= 0xFFFF0000u;
  return __builtin_bit_cast(float, u);
}

// one v_max3_f32 (plain fmaxf on MFMA outputs makes hipcc insert a canonicalising v_max per operand)
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <typename T, int DH> __device__ __forceinline__ void attn_body(const AttnParams& p) {
  using C = AttnCfg<T, DH>;
  constexpr bool BF = C::BF;
  constexpr int EPC = C::EPC, FS = C::FS, KROW = C::KROW, DB = C::DB, CPR = C::CPR, NLD = C::NLD;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q0 = blockIdx.x * 128 + wave * 32;
  const int q = q0 + l31;
  const bool q_ok = q < p.Nq;

  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.bsq + head * DH;
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.bsk + head * DH;
  const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.bsv + head * DH;
  T* O = reinterpret_cast<T*>(p.o) + (long)b * p.bso + head * DH;

  // ---- staging helpers ----
  uint4 kreg[NLD], vreg[NLD];
  unsigned short ones_val[NLD];
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int key = t0 + row;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (idx < 64 * CPR && key < p.Nk) {
        kv = *reinterpret_cast<const uint4*>(K + (long)key * p.ldk + ch * EPC);
        vv = *reinterpret_cast<const uint4*>(V + (long)key * p.ldv + ch * EPC);
      }
      kreg[i] = kv;
      vreg[i] = vv;
      ones_val[i] = (idx < 64 * CPR && key < p.Nk) ? (unsigned short)0x3F80 : (unsigned short)0;
    }
  };
  auto lstore = [&](int buf) {
    char* ks = smem + buf * C::TILE;
    char* vs = ks + C::K_BYTES;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + 256 * i;
      if (idx < 64 * CPR) {
        const int row = idx / CPR, ch = idx - row * CPR;
        *reinterpret_cast<uint4*>(ks + row * KROW + ch * 16) = kreg[i];
        if constexpr (BF) {
          *reinterpret_cast<uint4*>(vs + row * C::VROW_BF + ch * 16) = vreg[i];
          if constexpr (C::ONES)
            if (ch == 0)
              *reinterpret_cast<unsigned short*>(vs + row * C::VROW_BF + DH * 2) = ones_val[i];
        } else {
          Vec16<T> v;
          v.u = vreg[i];
#pragma unroll
          for (int e = 0; e < EPC; ++e)
            *reinterpret_cast<T*>(vs + (ch * EPC + e) * C::VROW_F32 + row * (int)sizeof(T)) = v.e[e];
        }
      }
    }
  };

  // the first K/V tile's global loads go out before anything waits (zero fill, Q fragments, the barrier below): the
  // short cross-attention launches (S = 77: two tiles per workgroup) are bound by exposed load latencies
  gload(0);

  // zero the whole LDS once: padding chunks (d >= DH) are never written again and must not hold NaNs
  for (int i = tid; i < C::LDS_BYTES / 16; i += 256) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);

  // Q fragments (B operand of S^T = K Q^T), pre-multiplied by scale*log2(e)
  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / (int)sizeof(T);
    Vec16<T> v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v.u = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
    if constexpr (C::MREF) {   // (the m_ref slot rides in the contraction: its scores must already be in the log2 domain)
#pragma unroll
      for (int e = 0; e < EPC; ++e) v.e[e] = from_f32<T>(to_f32<T>(v.e[e]) * sl2);
    }
    qf[s] = v.u;
  }

  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = C::MREF ? 0.f : -INFINITY, l_run = 0.f;  // MREF: m_run is the bf16-representable reference

  const int nt = (p.Nk + 63) / 64;
  __syncthreads();  // zero fill done
  if constexpr (C::MREF) {
    if (tid < 64 * C::NBUF)
      *reinterpret_cast<unsigned short*>(smem + (tid >> 6) * C::TILE + (tid & 63) * KROW + DH * 2) = 0x3F80;
  }
  lstore(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    const int buf = (C::NBUF == 2) ? (t & 1) : 0;
    if (t + 1 < nt) gload(t0 + 64);
    const char* ks = smem + buf * C::TILE;
    const char* vs = ks + C::K_BYTES;

    // ---- S^T = K Q^T : two 32-key blocks (scores already in the log2 domain) ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
#pragma unroll
      for (int st = 0; st < FS; ++st) {
        const uint4 a = *reinterpret_cast<const uint4*>(ks + (32 * kb + l31) * KROW + 32 * st + 16 * h);
        Mma<T>::step(a, qf[st], s[kb]);
      }
    }
    if (t0 + 64 > p.Nk) {  // partial last tile: mask keys >= Nk (wave-uniform branch)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + acc_row(r, h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    if (p.causal && t0 + 63 > q0) {  // causal mask: keys after the query (tiles wholly before this wave's queries skip it)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + acc_row(r, h) > q) s[kb][r] = -INFINITY;
    }
    // ---- online softmax (per lane = per query; the two lane halves hold disjoint keys) ----
    // tile maximum: four independent max3 chains, then across the two lane halves.  The chain heads read MFMA
    // results from inline asm: hipcc inserts no MFMA-result wait states for an asm statement (cdna guide 5.7 item 2),
    // and a too-early read returns stale registers (seen as run-to-run differences), so they sit behind an s_nop.
    // Wait states between an MFMA's result write and a VALU read of it that the compiler cannot see (cdna guide 5.7
    // item 2): 12 for the 8-pass v_mfma_f32_32x32x16_bf16 (s_nop 11 would do; s_nop 15 = 16 states is what ships);
    // the f32 parity build multiplies with the 16-pass v_mfma_f32_32x32x2_f32, whose result needs 8 more (about 20),
    // so it takes s_nop 15 + s_nop 7 = 24 states.  The nops are INSIDE the asm statement that reads the registers.
    float mxa, mxb, mxc, mxd;
#define AF_ATTN_MAX_HEADS(NOPS)                                                                                     \
    asm(NOPS "v_max3_f32 %0, %4, %5, %6\n\t"                                                                        \
             "v_max3_f32 %1, %7, %8, %9\n\t"                                                                        \
             "v_max3_f32 %2, %10, %11, %12\n\t"                                                                     \
             "v_max3_f32 %3, %13, %14, %15"                                                                         \
        : "=&v"(mxa), "=&v"(mxb), "=&v"(mxc), "=&v"(mxd)                                                            \
        : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][8]), "v"(s[0][9]), "v"(s[0][10]), "v"(s[1][0]),        \
          "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][8]), "v"(s[1][9]), "v"(s[1][10]))
    if constexpr (BF) AF_ATTN_MAX_HEADS("s_nop 15\n\t");
    else AF_ATTN_MAX_HEADS("s_nop 15\n\ts_nop 7\n\t");
#undef AF_ATTN_MAX_HEADS
    mxa = max3f(mxa, s[0][3], s[0][4]); mxb = max3f(mxb, s[0][11], s[0][12]);
    mxc = max3f(mxc, s[1][3], s[1][4]); mxd = max3f(mxd, s[1][11], s[1][12]);
    mxa = max3f(mxa, s[0][5], s[0][6]); mxb = max3f(mxb, s[0][13], s[0][14]);
    mxc = max3f(mxc, s[1][5], s[1][6]); mxd = max3f(mxd, s[1][13], s[1][14]);
    mxa = max3f(mxa, s[0][7], s[0][15]); mxc = max3f(mxc, s[1][7], s[1][15]);
    float mx = max3f(mxa, mxb, mxc);
    mx = xhalf_max(fmaxf(mx, mxd));
    if constexpr (C::MREF) {
      // s already holds score - m_ref (the MFMA added the Q-side "-m_ref" slot times the K-side 1.0), so the common path
      // is exp2(s) with no per-element subtraction.  The reference only moves on the first tile or when a score gets
      // more than 2^MREF_HEADROOM above it (fp32 / bf16 exponents have room; the final O / l ratio is scale free).
      constexpr float MREF_HEADROOM = 24.0f;
      const bool move = (t == 0) || mx > MREF_HEADROOM;
      if (__builtin_amdgcn_ballot_w64(move) != 0) {   // rare after the first tile
        const float m_new = move ? bf16_ceil(m_run + mx) : m_run;
        const float delta = m_new - m_run;  // exact: both are bf16 values
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
        if constexpr (!C::ONES) l_run *= alpha;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        m_run = m_new;
        if (h == C::MREF_HALF) {  // refresh the -m_ref slot of this lane's Q fragment
          Vec16<T> v;
          v.u = qf[C::MREF_STEP];
          v.e[C::MREF_ELEM] = from_f32<T>(-m_new);
          qf[C::MREF_STEP] = v.u;
        }
      }
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(s[kb][r]);
          s[kb][r] = pv;
          if constexpr (!C::ONES) psum += pv;
        }
      if constexpr (!C::ONES) l_run += psum;
    } else {
      // Q is NOT pre-scaled here (no second bf16 rounding of q): scale * log2(e) enters in the fp32 fma in front of the exponential
      const float m_new = fmaxf(m_run, mx);
      const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
      const float msl = m_new * sl2;
      float psum = 0.f;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float pv = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -msl));
          s[kb][r] = pv;
          if constexpr (!C::ONES) psum += pv;
        }
      if constexpr (!C::ONES) l_run = l_run * alpha + psum;
      m_run = m_new;
      if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // rescale only when some lane's max moved
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
      }
    }

    // ---- O^T += V^T P^T ----
    if constexpr (BF) {
      // lane -> address of its 4-element row piece for ds_read_b64_tr_b16 (16-lane groups, 4 rows x 16 cols)
      const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
      const char* vlane = vs + (4 * h + tq) * C::VROW_BF + (16 * gi + 4 * tp) * 2;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Vec16<T> pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) pb.e[j] = from_f32<T>(s[kb][8 * s2 + j]);
          const char* vrow = vlane + (32 * kb + 16 * s2) * C::VROW_BF;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            typedef s16x4 __attribute__((address_space(3))) * lds_v4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vrow + 64 * d));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vrow + 8 * C::VROW_BF + 64 * d));
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            const uint4 a = make_uint4(lo2.x, lo2.y, hi2.x, hi2.y);
            Mma<T>::step(a, pb.u, o[d]);
          }
        }
    } else {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          Vec16<T> pb;
#pragma unroll
          for (int e = 0; e < 4; ++e) pb.e[e] = from_f32<T>(s[kb][4 * g + e]);
          const int key0 = 32 * kb + 8 * g + 4 * h;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const uint4 a = *reinterpret_cast<const uint4*>(vs + (32 * d + l31) * C::VROW_F32 + key0 * 4);
            Mma<T>::step(a, pb.u, o[d]);
          }
        }
    }

    if (C::NBUF == 1) __syncthreads();  // single buffer: everyone has finished reading before it is rewritten
    if (t + 1 < nt) lstore((C::NBUF == 2) ? ((t + 1) & 1) : 0);
    __syncthreads();
  }

  float l_tot;
  if constexpr (C::ONES) {
    // row DH of O^T: block DH/32, in-block row rr -> register (rr&3)+4*(rr>>3) of lane-half (rr>>2)&1
    constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
    l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  } else {
    l_tot = l_run + __shfl_xor(l_run, 32, 64);
  }
  const float inv = 1.0f / l_tot;
  if (p.lse && q_ok && h == 0)
    p.lse[((long)blockIdx.z * p.H + blockIdx.y) * p.Nq + q] = (C::MREF ? m_run : m_run * sl2) + __builtin_amdgcn_logf(l_tot);
  if (q_ok) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * h;
        if (dd < DH) {
          Quad<T> ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(o[d][4 * g + e] * inv);
          ov.store(O + (long)q * p.ldo + dd);
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// dh = 40, bf16: the 64x64-level attention (self N = S = 4096: 14 % of a denoising step; cross S = 77).
//
// What the lab (scripts/lab/attn_lab.hip, interleaved A/B runs on one device) showed about the four-wave kernel above at
// this shape: with QK^T, softmax and PV removed it still took 55 % of its time — staging K/V through registers with a
// one-tile prefetch exposes the L2 latency (about 1 us under load) once per tile, and the 80-byte head slices of the
// [B][N][3C] activation cost 2.5x their size in cache-line traffic.  This kernel therefore
//   * shares every staged K/V tile among EIGHT waves (256 queries per workgroup, two workgroups per CU): half the
//     staging traffic and LDS writes per query;
//   * stages by LDS-DMA (buffer_load ... lds): no VGPR round trip, no ds_write, no "vmcnt(0) then store" in the middle
//     of each wave's instruction stream; tile t+1 is in flight while tile t is multiplied.  LDS-DMA writes lane-
//     linearly (wave-uniform base + 16 * lane), so padding and swizzle live in the per-lane SOURCE address and in the
//     EXEC mask (pad lanes are masked off):
//       K image: 64 rows x 112 B = 5 data chunks | [1.0, 0 x7] (the K side of the "-m_ref" slot, set once) | zeros
//       V image: 64 rows x 128 B = 5 data chunks | [1.0, 0 x7] (the ones column that sums the denominator) | 2 x zeros,
//                the two 64-byte halves of rows with bit 1 set swapped, so that the four rows a ds_read_b64_tr_b16
//                half touches cover all 64 banks (a plain 128-B stride would be 2-way);
//     rows >= Nk lie beyond the buffer resource's num_records and are zero-filled by the hardware range check;
//   * reads its fragments by inline asm with hand-counted lgkmcnt: hipcc puts s_waitcnt vmcnt(0) in front of every LDS
//     read it can see while an LDS-DMA is outstanding, which would expose the whole load latency each tile;
//   * takes the softmax reference m_ref from the FIRST tile only (it rides in the spare K slot of the QK^T MFMA, as
//     above) and runs no per-tile maximum afterwards: 18 v_max3 + the fix-up test per tile gone.  exp2(s - m_ref) can
//     then exceed 1, which is harmless (fp32 / bf16 exponents have the range; O / l is scale free) unless it overflows:
//     a score more than 2^127 above the first tile's maximum makes P, l or O non-finite — that, and only that, is
//     detected at the end (per lane, then per workgroup) and the workgroup repeats its block with the running-reference
//     loop of the kernel above.  tests/test_ops_gpu.py::test_attention_spiky_bf16_dh40 forces both paths.
// Measured (MI355X, lab, same random data, one process): 487-517 us -> 391-412 us for N = S = 4096, B = 16, H = 8.
// ---------------------------------------------------------------------------------------------------------------
namespace ring {
constexpr int NBUF = 2;
// DH = 40: the layout described above.  DH = 80 (round 4: the 32x32-level self-attention, N = S = 1024) is the same kernel
// with 10 data chunks per row: K rows of 11 chunks (176 B: 32 rows x 44 dwords are conflict-free for ds_read_b128; the
// sixth k step's upper half would be chunk 11 -- its Q half is zero, so those lanes read the m_ref chunk again), V rows of
// 12 chunks = three 64-byte d blocks (192 B: the four rows a ds_read_b64_tr_b16 half touches start at banks 0 / 48 / 32 /
// 16 -- no swap needed), 23 pieces per tile, three O^T blocks, 24 MFMAs per wave and tile; ~165 VGPRs: four-wave workgroups
// (W_ = 4: 128 queries share a staged tile), three per CU.
template <int DH_, int W_ = 8> struct Cfg {
  static constexpr int DH = DH_, NCH = DH / 8, WAVES = W_, NT = 64 * W_;   // W_ waves = 32 W_ queries share each staged tile                  // 16-byte data chunks per row
  static constexpr int FS = (DH + 1 + 15) / 16;                 // 16-wide k steps of QK^T, the m_ref slot included
  static constexpr int DB = (DH + 1 + 31) / 32;                 // 32-row blocks of O^T, the ones row included
  static constexpr int KCH = DH == 40 ? 7 : 11, VCH = 4 * DB;   // chunks per staged K / V row
  static constexpr int KROW = 16 * KCH, VROW = 16 * VCH;
  static constexpr int K_BYTES = 64 * KROW, V_BYTES = 64 * VROW, BUF = K_BYTES + V_BYTES;
  static constexpr int KP = KCH, VP = VCH, NPIECE = KP + VP;    // pieces of 1 KiB
  static constexpr int PPW = (NPIECE + WAVES - 1) / WAVES;
  static constexpr int LDS_BYTES = NBUF * BUF;    // dh 80: the pipelined pass keeps a three-deep ring
  static constexpr int MREF_STEP = DH / 16, MREF_HALF = (DH % 16) / 8, MREF_ELEM = 0;   // d = DH
  static constexpr bool VSWZ = VROW == 128;                     // 128-byte V rows: the halves of odd row pairs are swapped
  static constexpr bool KLASTFIX = 2 * (FS - 1) + 1 >= KCH;     // the last k step's upper half lies beyond the row
  static constexpr int KPADS = KCH - NCH, VPADS = VCH - NCH, PADS = 64 * (KPADS + VPADS);
  static_assert(DH % 8 == 0 && KCH > NCH && VCH > NCH && 2 * (FS - 1) < KCH, "row layout");
};
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (HIP's uint4 / uint2 are structs: asm operands must be vectors)

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
template <int OFF> __device__ __forceinline__ u32x2 lds_read_tr64(unsigned addr) {
  u32x2 v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
  return v;
}
// the wait names the registers it covers ("+v"): no consumer of them can be scheduled above it (cdna guide 5.7, form ii)
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
__device__ __forceinline__ void mma(const u32x4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// workgroup -> (query block, head, sample).  Workgroups go to the XCDs round-robin in launch order; mapped plainly, the query
// blocks of one head land on eight different XCDs and each XCD's L2 fetches that head's K / V for itself.  Here consecutive
// SLOTS of one XCD are the query blocks of one head: they run together on one L2 (heads x samples must divide by 8, else plain).
__device__ __forceinline__ void ring_coords(int& qb, int& head, int& b) {
  const int nqb = gridDim.x, H = gridDim.y, HB = gridDim.y * gridDim.z;
  if (HB & 7) { qb = blockIdx.x; head = blockIdx.y; b = blockIdx.z; return; }
  const int id = blockIdx.x + nqb * (blockIdx.y + H * blockIdx.z);
  const int xcd = id & 7, slot = id >> 3;
  const int grp = slot / nqb;
  qb = slot - grp * nqb;
  const int yz = grp * 8 + xcd;
  b = yz / H;
  head = yz - b * H;
}
template <int I, int N, typename F> __device__ __forceinline__ void sfor(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    sfor<I + 1, N>(f);
  }
}

// one pass over the keys for this workgroup's 256 queries.  RUNMAX = false: reference from the first tile only;
// returns whether this lane saw a non-finite result (outputs are written either way; a repeat overwrites them).
// MODE 0: softmax reference from the first tile only (RUNMAX = false above); 1: the running bf16 reference (the repeat);
// 2: the plain online softmax -- fp32 running row maximum, exp2(s - m), O rescaled when it moves -- with no m_ref slot in the
// contraction (one k step fewer when DH is a multiple of 16): operation for operation what attn_body does for dh 80, so the
// ring kernel is BIT-IDENTICAL to the four-wave kernel it replaces there (test_attention_dh80_ring_bit_identical).
template <int DH, int MODE, int W = 8> __device__ __forceinline__ bool pass(const AttnParams& p, char* smem) {
  typedef bf16 T;
  using C = Cfg<DH, W>;
  constexpr int WAVES = W, NT = 64 * W;
  constexpr bool RUNMAX = MODE == 1, EXACT = MODE == 2;
  constexpr int FS = EXACT ? (DH + 15) / 16 : C::FS, DB = C::DB, KROW = C::KROW, VROW = C::VROW, K_BYTES = C::K_BYTES, BUF = C::BUF, KP = C::KP,
                NPIECE = C::NPIECE, PPW = C::PPW;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int qb, head, b;
  ring_coords(qb, head, b);
  const int q = qb * (WAVES * 32) + wave * 32 + l31;
  const bool q_ok = q < p.Nq;
  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.bsq + head * DH;
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.bsk + head * DH;
  const T* V = reinterpret_cast<const T*>(p.v) + (long)b * p.bsv + head * DH;
  T* O = reinterpret_cast<T*>(p.o) + (long)b * p.bso + head * DH;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(K), 0, (p.Nk - 1) * p.ldk * 2 + DH * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(V), 0, (p.Nk - 1) * p.ldv * 2 + DH * 2, 0x00020000);

  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
  // ---- DMA pieces of this wave: piece = wave + 8 j; pieces 0..KP-1 = K image, the rest = V image.  Lane -> (row, chunk) of
  // the image; chunk >= NCH = pad (lane masked off) ----
  unsigned voff[PPW];
  bool act[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = wave + WAVES * j;
    if (piece < KP) {
      const int g = piece * 64 + lane, row = g / C::KCH, c = g - row * C::KCH;
      act[j] = c < C::NCH;
      voff[j] = (unsigned)(row * p.ldk * 2 + c * 16);
    } else {
      const int g = (piece - KP) * 64 + lane, row = g / C::VCH, pos = g - row * C::VCH;
      const int c = C::VSWZ ? pos ^ (((row >> 1) & 1) << 2) : pos;
      act[j] = c < C::NCH && piece < NPIECE;
      voff[j] = (unsigned)(row * p.ldv * 2 + c * 16);
    }
  }
  auto stage = [&](int buf, int t0) {
    char* base = smem + buf * BUF;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + WAVES * j;
      if (piece >= NPIECE) continue;
      // the tile base travels in the per-lane VECTOR offset: the buffer range check that zero-fills rows >= Nk of the
      // last tile is documented for voffset + instruction offset only, not for soffset
      if (piece < KP) { if (act[j]) lds_dma16(rs_k, base + piece * 1024, voff[j] + (unsigned)(t0 * p.ldk * 2), 0u); }
      else { if (act[j]) lds_dma16(rs_v, base + K_BYTES + (piece - KP) * 1024, voff[j] + (unsigned)(t0 * p.ldv * 2), 0u); }
    }
  };
  const int nt = (p.Nk + 63) / 64;
  stage(0, 0);
  // Q fragments (B operand of S^T = K Q^T), loaded BEHIND the first tile's LDS-DMA so that the two latencies overlap (round 4;
  // before, Q was loaded and scaled ahead of the staging); pre-multiplied by scale*log2(e) after the wait below
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / 2;
    qf[s] = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) qf[s] = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
  }
  // pads of both buffers (disjoint from everything the DMA writes; never written again): chunk NCH of a K row = [1.0, 0 x7]
  // (the K side of the -m_ref slot), of a V row the ones column; the chunks behind them zeros
  for (int idx = tid; idx < NBUF * C::PADS; idx += NT) {
    const int bufi = idx / C::PADS, r = idx - bufi * C::PADS;
    char* base = smem + bufi * BUF;
    if (r < 64 * C::KPADS) {
      const int row = r / C::KPADS, c = C::NCH + (r - row * C::KPADS);
      *reinterpret_cast<uint4*>(base + row * KROW + c * 16) = make_uint4(c == C::NCH ? 0x3F80u : 0u, 0, 0, 0);
    } else {
      const int rr2 = r - 64 * C::KPADS, row = rr2 / C::VPADS, c = C::NCH + (rr2 - row * C::VPADS);
      const int pos = C::VSWZ ? c ^ (((row >> 1) & 1) << 2) : c;
      *reinterpret_cast<uint4*>(base + K_BYTES + row * VROW + pos * 16) = make_uint4(c == C::NCH ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = EXACT ? -INFINITY : 0.f;
  // fragment read addresses (per lane): K row l31 at 16 h; V rows 4 h + tq, (swizzled) 64-byte block, 8 tp + 32 gi
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned kaddr0 = lds0 + (unsigned)(l31 * KROW + 16 * h);
  const unsigned kaddr_last = C::KLASTFIX ? lds0 + (unsigned)(l31 * KROW) : kaddr0;   // last k step: both halves read chunk 2 (FS - 1)
  const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
  const int vrow0 = 4 * h + tq, sw = C::VSWZ ? (vrow0 >> 1) & 1 : 0;
  const unsigned vaddr0 = lds0 + (unsigned)(K_BYTES + vrow0 * VROW + 32 * gi + 8 * tp);

  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own pieces of tile 0 landed, own pads written, Q here
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    Vec16<T> v;
    v.u = qf[s];
    if constexpr (!EXACT) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v.e[e] = from_f32<T>(to_f32<T>(v.e[e]) * sl2);
    }
    qf[s] = v.u;
  }

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    const unsigned bo = (unsigned)((t & 1) * BUF);
    if (t + 1 < nt) stage((t + 1) & 1, t0 + 64);   // its last readers passed the barrier that ended tile t-1
    // ---- S^T = K Q^T : two 32-key blocks (scores already in the log2 domain, minus m_ref through the spare slot) ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    {
      const unsigned ka = kaddr0 + bo, kl = kaddr_last + bo;
      u32x4 a[2][FS];
      sfor<0, 2 * FS>([&](auto mc) {
        constexpr int m = decltype(mc)::value, kb = m / FS, st = m % FS;
        a[kb][st] = (C::KLASTFIX && st == C::FS - 1) ? lds_read128<kb * 32 * KROW + 32 * st>(kl) : lds_read128<kb * 32 * KROW + 32 * st>(ka);
      });
      sfor<0, 2 * FS>([&](auto mc) {
        constexpr int m = decltype(mc)::value, kb = m / FS, st = m % FS;
        wait_lgkm<2 * FS - 1 - m>(a[kb][st]);
        mma(a[kb][st], qf[st], s[kb]);
      });
    }
    if (t0 + 64 > p.Nk) {  // partial last tile: mask keys >= Nk (wave-uniform branch)
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + acc_row(r, h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    if (EXACT || RUNMAX || t == 0) {
      // tile maximum (see the kernel above for the wait states in front of the asm reads of MFMA results)
      float mxa, mxb, mxc, mxd;
      asm("s_nop 15\n\t"
          "v_max3_f32 %0, %4, %5, %6\n\t"
          "v_max3_f32 %1, %7, %8, %9\n\t"
          "v_max3_f32 %2, %10, %11, %12\n\t"
          "v_max3_f32 %3, %13, %14, %15"
          : "=&v"(mxa), "=&v"(mxb), "=&v"(mxc), "=&v"(mxd)
          : "v"(s[0][0]), "v"(s[0][1]), "v"(s[0][2]), "v"(s[0][8]), "v"(s[0][9]), "v"(s[0][10]), "v"(s[1][0]),
            "v"(s[1][1]), "v"(s[1][2]), "v"(s[1][8]), "v"(s[1][9]), "v"(s[1][10]));
      mxa = max3f(mxa, s[0][3], s[0][4]); mxb = max3f(mxb, s[0][11], s[0][12]);
      mxc = max3f(mxc, s[1][3], s[1][4]); mxd = max3f(mxd, s[1][11], s[1][12]);
      mxa = max3f(mxa, s[0][5], s[0][6]); mxb = max3f(mxb, s[0][13], s[0][14]);
      mxc = max3f(mxc, s[1][5], s[1][6]); mxd = max3f(mxd, s[1][13], s[1][14]);
      mxa = max3f(mxa, s[0][7], s[0][15]); mxc = max3f(mxc, s[1][7], s[1][15]);
      float mx = max3f(mxa, mxb, mxc);
      mx = xhalf_max(fmaxf(mx, mxd));
      if constexpr (EXACT) {
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * sl2);
        const float msl = m_new * sl2;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[kb][r], sl2, -msl));
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {  // rescale only when some lane's maximum moved
#pragma unroll
          for (int d = 0; d < DB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        }
      }
      const bool move = !EXACT && ((t == 0) || mx > 24.0f);
      if constexpr (!EXACT)
      if (__builtin_amdgcn_ballot_w64(move) != 0) {
        const float m_new = move ? bf16_ceil(m_run + mx) : m_run;
        const float delta = m_new - m_run;  // exact: both are bf16 values
        const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] -= delta;
#pragma unroll
        for (int d = 0; d < DB; ++d)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        m_run = m_new;
        if (h == C::MREF_HALF) {  // refresh the -m_ref slot of this lane's Q fragment
          Vec16<T> v;
          v.u = qf[C::MREF_STEP];
          v.e[C::MREF_ELEM] = from_f32<T>(-m_new);
          qf[C::MREF_STEP] = v.u;
        }
      }
    }
    if constexpr (!EXACT) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    }
    // ---- O^T += V^T P^T : four k steps of 16 keys, DB blocks of 32 rows each ----
    sfor<0, 4>([&](auto kc) {
      constexpr int ks = decltype(kc)::value, KB = ks >> 1, S2 = ks & 1;
      constexpr int RO = (KB * 32 + S2 * 16) * VROW;
      u32x2 lo[DB], hi[DB];
      sfor<0, DB>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        const unsigned va = vaddr0 + bo + (unsigned)(64 * (d ^ sw));   // (sw = 0 unless the rows are 128 bytes: then DB = 2)
        lo[d] = lds_read_tr64<RO>(va);
        hi[d] = lds_read_tr64<RO + 8 * VROW>(va);
      });
      Vec16<T> pb;
#pragma unroll
      for (int j = 0; j < 8; ++j) pb.e[j] = from_f32<T>(s[KB][8 * S2 + j]);
      sfor<0, DB>([&](auto dc) {
        constexpr int d = decltype(dc)::value;
        wait_lgkm2<2 * (DB - 1 - d)>(lo[d], hi[d]);
        mma(u32x4{lo[d].x, lo[d].y, hi[d].x, hi[d].y}, pb.u, o[d]);
      });
    });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile t+1 landed
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone is done reading tile t
  }
  // row DH of O^T (the ones column): block DH/32, in-block row rr -> register (rr&3)+4*(rr>>3) of lane-half (rr>>2)&1
  constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
  const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  const float inv = 1.0f / l_tot;
  bool bad = !(l_tot > 0.f && l_tot < INFINITY);
  if (p.lse && q_ok && h == 0)
    p.lse[((long)b * p.H + head) * p.Nq + q] = (EXACT ? m_run * sl2 : m_run) + __builtin_amdgcn_logf(l_tot);
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int dd = 32 * d + 8 * g + 4 * h;
      if (dd < DH) {
        Quad<T> ov;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float val = o[d][4 * g + e] * inv;
          bad = bad || !(fabsf(val) < INFINITY);
          ov.e[e] = from_f32<T>(val);
        }
        if (q_ok) ov.store(O + (long)q * p.ldo + dd);
      }
    }
  return bad && q_ok;
}

template <int DH, int W = 8> __device__ __forceinline__ void ring_body(const AttnParams& p) {
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  if constexpr (DH == 80) {
    pass<DH, 2, W>(p, smem);       // exact running maximum: nothing can overflow, no repeat
  } else {
    const bool bad = pass<DH, 0, W>(p, smem);
    // (every DMA has been waited for and the last tile's barrier passed: LDS is free again)
    if (__syncthreads_or(bad ? 1 : 0)) pass<DH, 1, W>(p, smem);
  }
}
// dh 40: 128 VGPRs -> two workgroups per CU (four waves per SIMD)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_ring40_kernel(const AttnParams p) { ring_body<40>(p); }
// dh 80: four waves = 128 queries per workgroup, three workgroups per CU (<= 170 VGPRs): three waves per SIMD that do not share
// a barrier, and the plain online softmax (MODE 2: bit-identical to attn_kernel<bf16, 80>, so the bf16 forward does not move).
// Measured, N = S = 1024, B = 16, with the first-tile reference (MODE 0; 3 % more rms error at the op level than the exact
// maximum, whose largest P of a row is exactly 1.0): eight waves / two buffers / one workgroup per CU 63.2 us; eight waves with
// the S^T MFMAs of tile t + 1 issued between the exponentials of tile t and a four-deep ring behind counted vmcnt waits 64.5-66.2 us
// (in-kernel stamps: 17 % of a workgroup's cycles at the tile barrier -- the two waves of a SIMD reach their MFMA phases
// together -- and 18-25 % in the prologue burst); four waves x three workgroups 60.0 us; the four-wave register-staged kernel
// 79.8 us.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void attn_ring80_kernel(const AttnParams p) { ring_body<80, 4>(p); }
}  // namespace ring

// ---------------------------------------------------------------------------------------------------------------
// Cross-attention over a SHORT key list (the 77 context tokens; S <= 96), bf16, dh = 40 / 80 (the 64x64 and 32x32 levels).
//
// The flash kernels above spend a cross-attention launch in their prologue: per workgroup they stage two 64-key tiles,
// write pads, pass barriers and then run two tile iterations for 128-256 queries -- 43 / 28 us per launch for 0.4 GFLOP
// per sample against 42 / 21 MB of Q + O traffic (8 / 4 us at HBM speed).  With S <= 96 the whole K and V of a (sample,
// head) are 9-15 + 12-18 MFMA operand fragments: a wave keeps them in REGISTERS for its whole life and loops over blocks
// of 32 queries with no LDS and no barrier at all:
//     Q block (16-byte fragment loads, next block prefetched) -> S^T = K Q^T (3 key blocks) -> exact softmax over the <= 96
//     scores a lane holds (one cross-half maximum) -> P fed back as the B operand (accumulator-as-operand) ->
//     O^T = V^T P^T with the ones row giving the denominator -> 8-byte stores.
//   * K fragments come straight from the cached [B][S][2C] K|V tensor (lane = key, 16 bytes of the head's channels);
//   * V^T fragments come from a per-layer PACKED copy made once per af_set_context (pack_vt_kernel): element j of lane
//     (d, half h) of k-step s is V[key = 16 s + 8 (j >> 2) + 4 h + (j & 3)][d] -- the key order in which the S^T accumulator
//     comes back as an operand -- with row d = dh holding 1.0 (softmax denominator) and the other pad rows 0;
//   * a workgroup = 4 waves = 4 heads of one sample (neighbouring 80 / 160-byte slices of the same Q rows).
// ---------------------------------------------------------------------------------------------------------------
namespace xs {
constexpr int NKB = 3, SMAX = 32 * NKB;            // key blocks of 32; keys >= Nk are masked
template <int DH> struct Cfg {
  static constexpr int KS = (DH + 15) / 16;        // 16-wide k steps of QK^T (dh 40: 3, the last half zero)
  static constexpr int DB = (DH + 1 + 31) / 32;    // 32-row blocks of O^T including the ones row
  static constexpr int VSTEPS = 2 * NKB;           // 16-key k steps of PV
  static constexpr long PACK_ELEMS_PER_HEAD = (long)DB * VSTEPS * 2 * 32 * 8;
};

// vt[b][head][db][ks][h][l31][8]
template <int DH>
__global__ __launch_bounds__(256) void pack_vt_kernel(const bf16* __restrict__ v, int ldv, long bsv, int Nk, int H, int B,
                                                      bf16* __restrict__ vt) {
  using C = Cfg<DH>;
  const long total = (long)B * H * C::PACK_ELEMS_PER_HEAD;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int j = (int)(r & 7); r >>= 3;
    const int l31 = (int)(r & 31); r >>= 5;
    const int h = (int)(r & 1); r >>= 1;
    const int ks = (int)(r % C::VSTEPS); r /= C::VSTEPS;
    const int db = (int)(r % C::DB); r /= C::DB;
    const int head = (int)(r % H);
    const int b = (int)(r / H);
    const int d = db * 32 + l31;
    const int key = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
    float val = 0.f;
    if (d < DH) { if (key < Nk) val = (float)v[(long)b * bsv + (long)key * ldv + head * DH + d]; }
    else if (d == DH) val = 1.0f;
    vt[i] = (bf16)val;
  }
}

template <int DH>
__global__ __launch_bounds__(256) void xattn_short_kernel(const AttnParams p, const bf16* __restrict__ vt, int bpw) {
  using C = Cfg<DH>;
  typedef bf16 T;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y * 4 + wave, b = blockIdx.z;
  if (head >= p.H) return;                                          // (wave-uniform; no barrier in this kernel)
  const T* Q = reinterpret_cast<const T*>(p.q) + (long)b * p.bsq + head * DH;
  const T* K = reinterpret_cast<const T*>(p.k) + (long)b * p.bsk + head * DH;
  T* O = reinterpret_cast<T*>(p.o) + (long)b * p.bso + head * DH;
  const float sl2 = p.scale * 1.44269504088896340736f;

  // ---- resident operand fragments ----
  uint4 kf[NKB][C::KS];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      const int key = 32 * kb + l31, d0 = 16 * s + 8 * h;
      Vec16<T> v;
      v.u = make_uint4(0, 0, 0, 0);
      if (key < p.Nk && d0 < DH) v.u = *reinterpret_cast<const uint4*>(K + (long)key * p.ldk + d0);
      // K and Q go into the MFMA as loaded (round 4: scale * log2(e) used to ride on these fragments, a second bf16 rounding
      // of K); the factor enters in the fp32 fma in front of the exponential: softmax is exp2(s * sl2 - m * sl2)
      kf[kb][s] = v.u;
    }
  uint4 vf[C::DB][C::VSTEPS];
  {
    const uint4* vp = reinterpret_cast<const uint4*>(vt + ((long)b * p.H + head) * C::PACK_ELEMS_PER_HEAD);
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int ks = 0; ks < C::VSTEPS; ++ks) vf[db][ks] = vp[((db * C::VSTEPS + ks) * 2 + h) * 32 + l31];
  }
  auto load_q = [&](int blk, uint4 (&qf)[C::KS]) {
    const int q = blk * 32 + l31;
#pragma unroll
    for (int s = 0; s < C::KS; ++s) {
      const int d0 = 16 * s + 8 * h;
      qf[s] = make_uint4(0, 0, 0, 0);
      if (q < p.Nq && d0 < DH) qf[s] = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
    }
  };
  const int nblk = (p.Nq + 31) / 32;
  const int blk0 = blockIdx.x * bpw, blk1 = blk0 + bpw < nblk ? blk0 + bpw : nblk;
  if (blk0 >= nblk) return;
  // one block of 32 queries (fragments qraw, as loaded): S^T, softmax, O^T, store
  auto body = [&](const uint4 (&qf)[C::KS], int blk) {
    // ---- S^T = K Q^T ----
    f32x16 sc[NKB];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[kb][r] = 0.f;
#pragma unroll
      for (int s = 0; s < C::KS; ++s) Mma<T>::step(kf[kb][s], qf[s], sc[kb]);
    }
    // ---- exact softmax over the keys < Nk this lane's query column holds (rows split over the two lane halves) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
      if (32 * kb + 32 > p.Nk) {       // (wave-uniform) only a partial or empty key block has rows to mask
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * kb + acc_row(r, h) >= p.Nk) sc[kb][r] = -INFINITY;
      }
      // (plain fmaxf: an inline-asm v_max3 reading MFMA results would need its own wait states, see the flash kernels)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[kb][r]);
    }
    mx = xhalf_max(mx);
    const float msl = mx * sl2;
    uint4 pb[NKB][2];
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        Vec16<T> v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v.e[j] = from_f32<T>(__builtin_amdgcn_exp2f(fmaf(sc[kb][8 * s2 + j], sl2, -msl)));
        pb[kb][s2] = v.u;
      }
    // ---- O^T = V^T P^T (row DH = the softmax denominator) ----
    f32x16 o[C::DB];
#pragma unroll
    for (int db = 0; db < C::DB; ++db) {
#pragma unroll
      for (int r = 0; r < 16; ++r) o[db][r] = 0.f;
#pragma unroll
      for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) Mma<T>::step(vf[db][2 * kb + s2], pb[kb][s2], o[db]);
    }
    constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
    const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
    const float inv = 1.0f / l_tot;
    const int q = blk * 32 + l31;
#pragma unroll
    for (int db = 0; db < C::DB; ++db)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * db + 8 * g + 4 * h;
        if (dd < DH) {
          Quad<T> ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = from_f32<T>(o[db][4 * g + e] * inv);
          if (q < p.Nq) ov.store(O + (long)q * p.ldo + dd);
        }
      }
  };
  // Q fragments of THREE blocks in flight (a three-register ring, statically indexed): with one block ahead the wave had
  // ~3 KB of loads outstanding and the launch ran at the memory latency (1.3 TB/s of Q + O traffic), not at its bandwidth
  uint4 q0[C::KS], q1[C::KS], q2[C::KS];
  load_q(blk0, q0);
  load_q(blk0 + 1, q1);           // (blocks past the end load nothing: q >= Nq)
  load_q(blk0 + 2, q2);
  for (int blk = blk0; blk < blk1; blk += 3) {
    body(q0, blk);
    if (blk + 3 < blk1) load_q(blk + 3, q0);
    if (blk + 1 < blk1) {
      body(q1, blk + 1);
      if (blk + 4 < blk1) load_q(blk + 4, q1);
    }
    if (blk + 2 < blk1) {
      body(q2, blk + 2);
      if (blk + 5 < blk1) load_q(blk + 5, q2);
    }
  }
}
}  // namespace xs

template <typename T, int DH> __global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) { attn_body<T, DH>(p); }
// dh = 40 (the 64x64 self-attention, 16 % of a denoising step): 132 VGPRs as written; capped at 128 the kernel runs
// four waves per SIMD instead of three (LDS allows four workgroups per CU)
template <typename T, int DH>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_kernel_w4(const AttnParams p) {
  attn_body<T, DH>(p);
}

template <typename T, int DH> static int launch_attn(const AttnParams& p, int B, hipStream_t stream) {
  using C = AttnCfg<T, DH>;
  static unsigned long long attr_done = 0, attr_done_w4 = 0;
  if (int rc = af_ensure_dynamic_lds(attr_done, reinterpret_cast<const void*>(&attn_kernel<T, DH>), C::LDS_BYTES)) return rc;
  if (int rc = af_ensure_dynamic_lds(attr_done_w4, reinterpret_cast<const void*>(&attn_kernel_w4<T, DH>), C::LDS_BYTES)) return rc;
  if constexpr (C::BF && DH == 40) {
    if ((g_af_knobs.attn_ring & 1) && !p.causal) {   // eight-wave LDS-DMA ring kernel (the 64x64 level)
      static unsigned long long attr_done_ring = 0;
      if (int rc = af_ensure_dynamic_lds(attr_done_ring, reinterpret_cast<const void*>(&ring::attn_ring40_kernel), ring::Cfg<40>::LDS_BYTES)) return rc;
      hipLaunchKernelGGL(ring::attn_ring40_kernel, dim3((p.Nq + 255) / 256, p.H, B), dim3(512), ring::Cfg<40>::LDS_BYTES, stream, p);
      HIP_CHECK_RET(hipGetLastError());
      return 0;
    }
  }
  if constexpr (C::BF && DH == 80) {
    if ((g_af_knobs.attn_ring & 2) && !p.causal && (p.Nk >= 256 || (g_af_knobs.attn_ring & 4))) {   // 80-wide heads (the 32x32 level; bit 2: tests force it for short key lists)
      static unsigned long long attr_done_ring = 0;
      constexpr int lds80 = ring::Cfg<80, 4>::LDS_BYTES;
      if (int rc = af_ensure_dynamic_lds(attr_done_ring, reinterpret_cast<const void*>(&ring::attn_ring80_kernel), lds80)) return rc;
      hipLaunchKernelGGL(ring::attn_ring80_kernel, dim3((p.Nq + 127) / 128, p.H, B), dim3(256), lds80, stream, p);
      HIP_CHECK_RET(hipGetLastError());
      return 0;
    }
  }
  dim3 grid((p.Nq + 127) / 128, p.H, B);
  if (C::BF && DH == 40)
    hipLaunchKernelGGL((attn_kernel_w4<T, DH>), grid, dim3(256), C::LDS_BYTES, stream, p);
  else
    hipLaunchKernelGGL((attn_kernel<T, DH>), grid, dim3(256), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
  return 0;
}

// packed V^T fragments for xs::xattn_short_kernel (elements; 0 = this (dtype, dh, Nk) has no short-key kernel)
template <typename T> long af_attn_short_pack_elems(int B, int H, int dh, int Nk) {
  if (sizeof(T) != 2 || Nk <= 0 || Nk > xs::SMAX) return 0;
  if (dh == 40) return (long)B * H * xs::Cfg<40>::PACK_ELEMS_PER_HEAD;
  if (dh == 80) return (long)B * H * xs::Cfg<80>::PACK_ELEMS_PER_HEAD;
  return 0;
}
template <typename T>
int af_launch_attn_short_pack(const void* v, int ldv, long bsv, int Nk, int H, int dh, int B, void* vt, hipStream_t stream) {
  if constexpr (sizeof(T) == 2) {
    const long total = af_attn_short_pack_elems<T>(B, H, dh, Nk);
    if (total <= 0) { af_set_error_msg("attention: no short-key pack for dh %d, %d keys", dh, Nk); return -1; }
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    if (dh == 40)
      hipLaunchKernelGGL((xs::pack_vt_kernel<40>), dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16*>(v), ldv, bsv, Nk, H, B, reinterpret_cast<bf16*>(vt));
    else
      hipLaunchKernelGGL((xs::pack_vt_kernel<80>), dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const bf16*>(v), ldv, bsv, Nk, H, B, reinterpret_cast<bf16*>(vt));
    HIP_CHECK_RET(hipGetLastError());
    return 0;
  } else {
    af_set_error_msg("attention: the short-key kernel is bf16 only");
    return -1;
  }
}
std::atomic<long> g_af_attn_short_launches{0};

template <int DH> static int launch_xattn_short(const AttnParams& p, int B, hipStream_t stream) {
  const int nblk = (p.Nq + 31) / 32;
  // blocks of 32 queries per wave: the launch should fit the chip in ONE round of workgroups (every wave pays the load of its
  // resident K / V^T fragments): 1024 SIMDs x the waves per SIMD the kernel's registers allow (dh 40: two, dh 80: one)
  const long wave_blocks = (long)B * p.H * nblk;
  const long slots = 1024L * (DH == 40 ? 2 : 1);
  int bpw = (int)((wave_blocks + slots - 1) / slots);
  if (bpw < 1) bpw = 1;
  if (bpw > 16) bpw = 16;
  dim3 grid((nblk + bpw - 1) / bpw, (p.H + 3) / 4, B);
  hipLaunchKernelGGL((xs::xattn_short_kernel<DH>), grid, dim3(256), 0, stream, p, reinterpret_cast<const bf16*>(p.vt_pack), bpw);
  HIP_CHECK_RET(hipGetLastError());
  ++g_af_attn_short_launches;
  return 0;
}

template <typename T> int af_launch_attention(const AttnParams& p, int B, int dh, hipStream_t stream) {
  constexpr int EPC = 16 / sizeof(T);
  if (p.ldq % EPC || p.ldk % EPC || p.ldv % EPC || p.ldo % 4 || p.Nk <= 0) {
    af_set_error_msg("attention: row strides must be multiples of %d elements", EPC);
    return -1;
  }
  if (p.Nq <= 0 || B <= 0) return 0;
  AfProfScope prof(AF_K_ATTENTION, stream, 4.0 * B * p.H * (double)p.Nq * p.Nk * dh,
                   (2.0 * p.Nq + 2.0 * p.Nk) * B * p.H * dh * sizeof(T));
  if constexpr (sizeof(T) == 2) {
    // short key list with the V^T fragments packed by the caller: the register-resident cross-attention kernel
    if (p.vt_pack && g_af_knobs.attn_short && !p.causal && !p.lse && p.Nk <= xs::SMAX && (dh == 40 || dh == 80))
      return dh == 40 ? launch_xattn_short<40>(p, B, stream) : launch_xattn_short<80>(p, B, stream);
  }
  switch (dh) {
    case 8: return launch_attn<T, 8>(p, B, stream);
    case 16: return launch_attn<T, 16>(p, B, stream);
    case 32: return launch_attn<T, 32>(p, B, stream);
    case 40: return launch_attn<T, 40>(p, B, stream);
    case 64: return launch_attn<T, 64>(p, B, stream);
    case 80: return launch_attn<T, 80>(p, B, stream);
    case 128: return launch_attn<T, 128>(p, B, stream);
    case 160: return launch_attn<T, 160>(p, B, stream);
    default:
      af_set_error_msg("attention: unsupported head dim %d", dh);
      return -1;
  }
}

template int af_launch_attention<bf16>(const AttnParams&, int, int, hipStream_t);
template long af_attn_short_pack_elems<bf16>(int, int, int, int);
template long af_attn_short_pack_elems<float>(int, int, int, int);
template int af_launch_attn_short_pack<bf16>(const void*, int, long, int, int, int, int, void*, hipStream_t);
template int af_launch_attn_short_pack<float>(const void*, int, long, int, int, int, int, void*, hipStream_t);
template int af_launch_attention<float>(const AttnParams&, int, int, hipStream_t);
