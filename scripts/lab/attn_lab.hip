// Standalone lab for the dh = 40 self-attention kernel (N = S = 4096, 8 heads, batch 16: 14 % of a denoising step).
// Not part of the shipped library.  Build (here) and run (GPU box):
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -o scripts/lab/attn_lab scripts/lab/attn_lab.hip
//   scripts/lab/attn_lab [reps]
// Variants are template flags of ONE kernel body so that ablations (wrong results, timing only) and candidate
// structures run interleaved in one process on the same random data (cdna guide rules 17, 24, 25).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct P {
  const bf16 *q, *k, *v;
  bf16* o;
  int ldq, ldk, ldv, ldo;
  long bsq, bsk, bsv, bso;
  long hsq, hsk, hsv;   // head strides (elements): DH for the interleaved [B][N][H*dh] layout, B*N*dh for head-major
  int Nq, Nk, H;
  float scale;
};

__device__ __forceinline__ float xhalf_max(float v) {   // (inline asm: hipcc folds fmaxf over the builtin's two results)
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__device__ __forceinline__ float bf16_ceil(float x) {
  unsigned u = __builtin_bit_cast(unsigned, x);
  if (x > 0.f && (u & 0xFFFFu)) u += 0x10000u;
  u &= 0xFFFF0000u;
  return __builtin_bit_cast(float, u);
}
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
__device__ __forceinline__ void mma(const uint4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
union V16 { uint4 u; bf16 e[8]; };
union Q4 { uint2 u; bf16 e[4]; };

// FLAGS: 1 NOMAX (reference from tile 0 only, no max chain afterwards; overflow is NOT handled here)
//        2 XCD   (the 32 query blocks of one (batch, head) run on one XCD: K/V fetched into one L2)
//        4 ABL_NOEXP   8 ABL_NOPV   16 ABL_NOQK   32 ABL_NOSTAGE (no global loads / LDS stores after tile 0)   (timing only)
//        64 NOBAR2 (double buffer: one barrier per tile is already the case; reserved)
constexpr int DH = 40, FS = 3, KROW = FS * 32 + 16, DB = 2, VROW = 192, CPR = 5;
constexpr int K_BYTES = 64 * KROW, V_BYTES = 64 * VROW, TILE = K_BYTES + V_BYTES;
constexpr int MREF_STEP = 2, MREF_HALF = 1, MREF_ELEM = 0;   // d = 40 -> byte 80 = step 2 (64..95), half 1 (80..95), elem 0

template <int FLAGS, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4)))
void attn_lab(const P p) {
  constexpr bool NOMAX = FLAGS & 1, XCD = FLAGS & 2, NOEXP = FLAGS & 4, NOPV = FLAGS & 8, NOQK = FLAGS & 16, NOSTAGE = FLAGS & 32;
  constexpr int NT = WAVES * 64;
  constexpr int NLD = (64 * CPR + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = lane >> 5, l31 = lane & 31;
  int bx = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if constexpr (XCD) {
    // linear id in dispatch order; ids i, i+8, ... share an XCD.  Give XCD x the (batch, head) pairs x, x+8, ...:
    // all query blocks of a pair then sit on one XCD.
    const int nqb = gridDim.x, id = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, slot = id >> 3;              // slot-th workgroup of this XCD
    const int pair = (slot / nqb) * 8 + xcd;             // (batch, head) pair index; gridDim.y * gridDim.z % 8 == 0
    bx = slot % nqb;
    head = pair % gridDim.y;
    b = pair / gridDim.y;
  }
  const int q0 = bx * (WAVES * 32) + wave * 32;
  const int q = q0 + l31;
  const bool q_ok = q < p.Nq;
  const bf16* Q = p.q + (long)b * p.bsq + head * p.hsq;
  const bf16* K = p.k + (long)b * p.bsk + head * p.hsk;
  const bf16* V = p.v + (long)b * p.bsv + head * p.hsv;
  bf16* O = p.o + (long)b * p.bso + head * DH;

  uint4 kreg[NLD], vreg[NLD];
  unsigned short ones_val[NLD];
  auto gload = [&](int t0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + NT * i;
      const int row = idx / CPR, ch = idx - row * CPR;
      const int key = t0 + row;
      uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
      if (idx < 64 * CPR && key < p.Nk) {
        kv = *reinterpret_cast<const uint4*>(K + (long)key * p.ldk + ch * 8);
        vv = *reinterpret_cast<const uint4*>(V + (long)key * p.ldv + ch * 8);
      }
      kreg[i] = kv;
      vreg[i] = vv;
      ones_val[i] = (idx < 64 * CPR && key < p.Nk) ? (unsigned short)0x3F80 : (unsigned short)0;
    }
  };
  auto lstore = [&](int buf) {
    char* ks = smem + buf * TILE;
    char* vs = ks + K_BYTES;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int idx = tid + NT * i;
      if (idx < 64 * CPR) {
        const int row = idx / CPR, ch = idx - row * CPR;
        *reinterpret_cast<uint4*>(ks + row * KROW + ch * 16) = kreg[i];
        *reinterpret_cast<uint4*>(vs + row * VROW + ch * 16) = vreg[i];
        if (ch == 0) *reinterpret_cast<unsigned short*>(vs + row * VROW + DH * 2) = ones_val[i];
      }
    }
  };
  gload(0);
  for (int i = tid; i < 2 * TILE / 16; i += NT) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / 2;
    V16 v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v.u = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v.e[e] = (bf16)((float)v.e[e] * sl2);
    qf[s] = v.u;
  }
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = 0.f;
  const int nt = (p.Nk + 63) / 64;
  __syncthreads();
  if (tid < 128) *reinterpret_cast<unsigned short*>(smem + (tid >> 6) * TILE + (tid & 63) * KROW + DH * 2) = 0x3F80;
  lstore(0);
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    const int buf = t & 1;
    if (t + 1 < nt && !(NOSTAGE && t > 0)) gload(t0 + 64);
    const char* ks = smem + buf * TILE;
    const char* vs = ks + K_BYTES;
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
      if (!NOQK || t == 0) {
#pragma unroll
        for (int st = 0; st < FS; ++st) {
          const uint4 a = *reinterpret_cast<const uint4*>(ks + (32 * kb + l31) * KROW + 32 * st + 16 * h);
          mma(a, qf[st], s[kb]);
        }
      }
    }
    if (t0 + 64 > p.Nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + ((r & 3) + 8 * (r >> 2) + 4 * h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    bool do_max = !NOMAX || t == 0;
    if (do_max) {
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
      const bool move = (t == 0) || mx > 24.0f;
      if (__builtin_amdgcn_ballot_w64(move) != 0) {
        const float m_new = move ? bf16_ceil(m_run + mx) : m_run;
        const float delta = m_new - m_run;
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
        if (h == MREF_HALF) {
          V16 v;
          v.u = qf[MREF_STEP];
          v.e[MREF_ELEM] = (bf16)(-m_new);
          qf[MREF_STEP] = v.u;
        }
      }
    }
    if (!NOEXP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    }
    if (!NOPV || t == 0) {
      const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
      const char* vlane = vs + (4 * h + tq) * VROW + (16 * gi + 4 * tp) * 2;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          V16 pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) pb.e[j] = (bf16)s[kb][8 * s2 + j];
          const char* vrow = vlane + (32 * kb + 16 * s2) * VROW;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            typedef s16x4 __attribute__((address_space(3))) * lds_v4;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vrow + 64 * d));
            const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(vrow + 8 * VROW + 64 * d));
            const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
            mma(make_uint4(lo2.x, lo2.y, hi2.x, hi2.y), pb.u, o[d]);
          }
        }
    } else {
      asm volatile("" ::"v"(s[0][0]), "v"(s[0][5]), "v"(s[1][3]), "v"(s[1][15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][r] += s[0][r] + s[1][r];
    }
    if (t + 1 < nt && !(NOSTAGE && t > 0)) lstore((t + 1) & 1);
    __syncthreads();
  }
  constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
  const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * h;
        if (dd < DH) {
          Q4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = (bf16)(o[d][4 * g + e] * inv);
          *reinterpret_cast<uint2*>(O + (long)q * p.ldo + dd) = ov.u;
        }
      }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// LDS-DMA staged variant.  K/V tiles go HBM/L2 -> LDS by buffer_load ... lds (no VGPR round trip, no ds_write, no
// "vmcnt(0) then store" in the middle of every wave's instruction stream); tile t+1 is in flight while tile t is
// multiplied.  LDS-DMA writes lane-linearly (wave-uniform base + 16 * lane), so padding and swizzle live in the per-lane
// SOURCE address and in the EXEC mask:
//   K image: 64 rows x 112 B (7 chunks: 5 data | [1.0, 0 x7] = the K side of the "-m_ref" slot | zeros).  Only the data
//            chunks are written by the DMA (pad lanes are masked off); the pads are set once per buffer at kernel start.
//   V image: 64 rows x 128 B (8 chunks: 5 data | [1.0, 0 x7] = the ones column that sums the softmax denominator |
//            2 x zeros), the two 64-byte halves of rows with bit 1 set are swapped so that the four rows a
//            ds_read_b64_tr_b16 half touches cover all 64 banks (a plain 128-B stride would be 2-way).
// Fragment reads are inline asm (hipcc puts s_waitcnt vmcnt(0) in front of every LDS read it can see while an LDS-DMA
// is outstanding, which would expose the whole load latency each tile); waits are counted by hand.
// FLAGS: 1 NOMAX, 4 NOEXP, 8 NOPV, 16 NOQK (timing only)
// ---------------------------------------------------------------------------------------------------------------
constexpr int KROW2 = 112, VROW2 = 128;
constexpr int K2_BYTES = 64 * KROW2, V2_BYTES = 64 * VROW2, BUF2 = K2_BYTES + V2_BYTES;   // 7168 + 8192 = 15360
constexpr int KP = K2_BYTES / 1024, VP = V2_BYTES / 1024, NPIECE = KP + VP;                // 7 + 8 pieces of 1 KiB

__device__ __forceinline__ void lds_dma16(__amdgpu_buffer_rsrc_t rs, char* lds_base, unsigned voffset, unsigned soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_base, 16, voffset, soffset, 0, 0);
#endif
}
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;   // (HIP's uint4 / uint2 are structs: as asm operands they go through memory)
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
template <int N> __device__ __forceinline__ void wait_lgkm(u32x4& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory"); }
__device__ __forceinline__ void mma(const u32x4& a, const uint4& b, f32x16& c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int FLAGS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_dma(const P p) {
  constexpr bool NOMAX = FLAGS & 1, NOEXP = FLAGS & 4, NOPV = FLAGS & 8, NOQK = FLAGS & 16;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 128 + wave * 32 + l31;
  const bool q_ok = q < p.Nq;
  const bf16* Q = p.q + (long)b * p.bsq + head * p.hsq;
  const bf16* K = p.k + (long)b * p.bsk + head * p.hsk;
  const bf16* V = p.v + (long)b * p.bsv + head * p.hsv;
  bf16* O = p.o + (long)b * p.bso + head * DH;
  // rows >= Nk start beyond num_records: the hardware range check zero-fills them (tails, S = 77)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(K), 0, (p.Nk - 1) * p.ldk * 2 + DH * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(V), 0, (p.Nk - 1) * p.ldv * 2 + DH * 2, 0x00020000);

  // ---- DMA pieces of this wave: piece = wave + 4 j (j < 4); pieces 0..6 = K image, 7..14 = V image ----
  unsigned voff[4];
  bool act[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = wave + 4 * j;
    if (piece < KP) {
      const int g = piece * 64 + lane, row = g / 7, c = g - row * 7;
      act[j] = c < 5;
      voff[j] = (unsigned)(row * p.ldk * 2 + c * 16);
    } else {
      const int g = (piece - KP) * 64 + lane, row = g >> 3, c = (g & 7) ^ (((row >> 1) & 1) << 2);
      act[j] = c < 5 && piece < NPIECE;
      voff[j] = (unsigned)(row * p.ldv * 2 + c * 16);
    }
  }
  auto stage = [&](int buf, int t0) {
    char* base = smem + buf * BUF2;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int piece = wave + 4 * j;
      if (piece >= NPIECE) continue;
      if (piece < KP) { if (act[j]) lds_dma16(rs_k, base + piece * 1024, voff[j], (unsigned)(t0 * p.ldk * 2)); }
      else { if (act[j]) lds_dma16(rs_v, base + K2_BYTES + (piece - KP) * 1024, voff[j], (unsigned)(t0 * p.ldv * 2)); }
    }
  };
  stage(0, 0);
  // pads of both buffers (disjoint from everything the DMA writes)
  for (int idx = tid; idx < 2 * 320; idx += 256) {
    const int bufi = idx / 320, r = idx - bufi * 320;
    char* base = smem + bufi * BUF2;
    if (r < 128) {
      const int row = r >> 1, c = 5 + (r & 1);
      *reinterpret_cast<uint4*>(base + row * KROW2 + c * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    } else {
      const int rr2 = r - 128, row = rr2 / 3, c = 5 + (rr2 - row * 3);
      const int pos = c ^ (((row >> 1) & 1) << 2);
      *reinterpret_cast<uint4*>(base + K2_BYTES + row * VROW2 + pos * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / 2;
    V16 v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v.u = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v.e[e] = (bf16)((float)v.e[e] * sl2);
    qf[s] = v.u;
  }
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = 0.f;
  const int nt = (p.Nk + 63) / 64;
  // fragment read addresses (per lane): K row l31 at 16 h; V rows 4 h + tq, swizzled 64-byte half, 8 tp + 32 gi
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned kaddr0 = lds0 + (unsigned)(l31 * KROW2 + 16 * h);
  const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
  const int vrow0 = 4 * h + tq, sw = (vrow0 >> 1) & 1;
  const unsigned vaddr0 = lds0 + (unsigned)(K2_BYTES + vrow0 * VROW2 + 32 * gi + 8 * tp);
  const unsigned vhalf[2] = {(unsigned)(64 * (0 ^ sw)), (unsigned)(64 * (1 ^ sw))};

  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile 0 (and the Q loads) have landed
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    const unsigned bo = (unsigned)((t & 1) * BUF2);
    if (t + 1 < nt) stage((t + 1) & 1, t0 + 64);
    // ---- S^T = K Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if (!NOQK || t == 0) {
      const unsigned ka = kaddr0 + bo;
      u32x4 a00 = lds_read128<0 * 3584 + 0>(ka), a01 = lds_read128<0 * 3584 + 32>(ka), a02 = lds_read128<0 * 3584 + 64>(ka);
      u32x4 a10 = lds_read128<1 * 3584 + 0>(ka), a11 = lds_read128<1 * 3584 + 32>(ka), a12 = lds_read128<1 * 3584 + 64>(ka);
      wait_lgkm<5>(a00); mma(a00, qf[0], s[0]);
      wait_lgkm<4>(a01); mma(a01, qf[1], s[0]);
      wait_lgkm<3>(a02); mma(a02, qf[2], s[0]);
      wait_lgkm<2>(a10); mma(a10, qf[0], s[1]);
      wait_lgkm<1>(a11); mma(a11, qf[1], s[1]);
      wait_lgkm<0>(a12); mma(a12, qf[2], s[1]);
    }
    if (t0 + 64 > p.Nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + ((r & 3) + 8 * (r >> 2) + 4 * h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    if (!NOMAX || t == 0) {
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
      const bool move = (t == 0) || mx > 24.0f;
      if (__builtin_amdgcn_ballot_w64(move) != 0) {
        const float m_new = move ? bf16_ceil(m_run + mx) : m_run;
        const float delta = m_new - m_run;
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
        if (h == MREF_HALF) {
          V16 v;
          v.u = qf[MREF_STEP];
          v.e[MREF_ELEM] = (bf16)(-m_new);
          qf[MREF_STEP] = v.u;
        }
      }
    }
    if (!NOEXP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    }
    // ---- O^T += V^T P^T ----
    if (!NOPV || t == 0) {
      const unsigned va0 = vaddr0 + bo + vhalf[0], va1 = vaddr0 + bo + vhalf[1];
#define AF_PV_STEP(KB, S2)                                                                                           \
      {                                                                                                              \
        constexpr int RO = ((KB) * 32 + (S2) * 16) * VROW2;                                                          \
        u32x2 lo0 = lds_read_tr64<RO>(va0), hi0 = lds_read_tr64<RO + 8 * VROW2>(va0);                                \
        u32x2 lo1 = lds_read_tr64<RO>(va1), hi1 = lds_read_tr64<RO + 8 * VROW2>(va1);                                \
        V16 pb;                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) pb.e[j] = (bf16)s[KB][8 * (S2) + j];                           \
        wait_lgkm2<2>(lo0, hi0);                                                                                     \
        mma(u32x4{lo0.x, lo0.y, hi0.x, hi0.y}, pb.u, o[0]);                                                     \
        wait_lgkm2<0>(lo1, hi1);                                                                                     \
        mma(u32x4{lo1.x, lo1.y, hi1.x, hi1.y}, pb.u, o[1]);                                                     \
      }
      AF_PV_STEP(0, 0) AF_PV_STEP(0, 1) AF_PV_STEP(1, 0) AF_PV_STEP(1, 1)
#undef AF_PV_STEP
    } else {
      asm volatile("" ::"v"(s[0][0]), "v"(s[0][5]), "v"(s[1][3]), "v"(s[1][15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][r] += s[0][r] + s[1][r];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // own pieces of tile t+1 landed
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone is done reading tile t
  }
  constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
  const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * h;
        if (dd < DH) {
          Q4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = (bf16)(o[d][4 * g + e] * inv);
          *reinterpret_cast<uint2*>(O + (long)q * p.ldo + dd) = ov.u;
        }
      }
  }
}
// Ring variant: WAVES x 32 queries per workgroup share each staged K/V tile; NBUF LDS buffers, tiles t+1 .. t+NBUF-1 in
// flight while tile t is multiplied (counted vmcnt, raw barrier).
template <int FLAGS, int WAVES, int NBUF>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(4, 4))) void attn_ring(const P p) {
  constexpr int NT = WAVES * 64, PPW = (NPIECE + WAVES - 1) / WAVES, D = NBUF - 1;
  constexpr bool NOMAX = FLAGS & 1, NOEXP = FLAGS & 4, NOPV = FLAGS & 8, NOQK = FLAGS & 16;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int h = lane >> 5, l31 = lane & 31;
  int bx = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
  if constexpr ((FLAGS & 2) != 0) {   // all query blocks of a (batch, head) pair on one XCD
    const int nqb = gridDim.x, id = blockIdx.x + nqb * (blockIdx.y + gridDim.y * blockIdx.z);
    const int xcd = id & 7, slot = id >> 3;
    const int pair = (slot / nqb) * 8 + xcd;
    bx = slot % nqb;
    head = pair % gridDim.y;
    b = pair / gridDim.y;
  }
  const int q = bx * (WAVES * 32) + wave * 32 + l31;
  const bool q_ok = q < p.Nq;
  const bf16* Q = p.q + (long)b * p.bsq + head * p.hsq;
  const bf16* K = p.k + (long)b * p.bsk + head * p.hsk;
  const bf16* V = p.v + (long)b * p.bsv + head * p.hsv;
  bf16* O = p.o + (long)b * p.bso + head * DH;
  // rows >= Nk start beyond num_records: the hardware range check zero-fills them (tails, S = 77)
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(K), 0, (p.Nk - 1) * p.ldk * 2 + DH * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(V), 0, (p.Nk - 1) * p.ldv * 2 + DH * 2, 0x00020000);

  // ---- DMA pieces of this wave: piece = wave + 4 j (j < 4); pieces 0..6 = K image, 7..14 = V image ----
  unsigned voff[PPW];
  bool act[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = wave + WAVES * j;
    if (piece < KP) {
      const int g = piece * 64 + lane, row = g / 7, c = g - row * 7;
      act[j] = c < 5;
      voff[j] = (unsigned)(row * p.ldk * 2 + c * 16);
    } else {
      const int g = (piece - KP) * 64 + lane, row = g >> 3, c = (g & 7) ^ (((row >> 1) & 1) << 2);
      act[j] = c < 5 && piece < NPIECE;
      voff[j] = (unsigned)(row * p.ldv * 2 + c * 16);
    }
  }
  auto stage = [&](int buf, int t0) {
    char* base = smem + buf * BUF2;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + WAVES * j;
      if (piece >= NPIECE) continue;
      if (piece < KP) { if (act[j]) lds_dma16(rs_k, base + piece * 1024, voff[j], (unsigned)(t0 * p.ldk * 2)); }
      else { if (act[j]) lds_dma16(rs_v, base + K2_BYTES + (piece - KP) * 1024, voff[j], (unsigned)(t0 * p.ldv * 2)); }
    }
  };
  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / 2;
    V16 v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v.u = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v.e[e] = (bf16)((float)v.e[e] * sl2);
    qf[s] = v.u;
  }
  const int nt = (p.Nk + 63) / 64;
#pragma unroll
  for (int d = 0; d < D; ++d)
    if (d < nt) stage(d, d * 64);
  // pads of every buffer (disjoint from everything the DMA writes)
  for (int idx = tid; idx < NBUF * 320; idx += NT) {
    const int bufi = idx / 320, r = idx - bufi * 320;
    char* base = smem + bufi * BUF2;
    if (r < 128) {
      const int row = r >> 1, c = 5 + (r & 1);
      *reinterpret_cast<uint4*>(base + row * KROW2 + c * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    } else {
      const int rr2 = r - 128, row = rr2 / 3, c = 5 + (rr2 - row * 3);
      const int pos = c ^ (((row >> 1) & 1) << 2);
      *reinterpret_cast<uint4*>(base + K2_BYTES + row * VROW2 + pos * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = 0.f;
  // fragment read addresses (per lane): K row l31 at 16 h; V rows 4 h + tq, swizzled 64-byte half, 8 tp + 32 gi
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned kaddr0 = lds0 + (unsigned)(l31 * KROW2 + 16 * h);
  const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
  const int vrow0 = 4 * h + tq, sw = (vrow0 >> 1) & 1;
  const unsigned vaddr0 = lds0 + (unsigned)(K2_BYTES + vrow0 * VROW2 + 32 * gi + 8 * tp);
  const unsigned vhalf[2] = {(unsigned)(64 * (0 ^ sw)), (unsigned)(64 * (1 ^ sw))};

  // own pieces of tile 0 (and the Q loads, which are older) have landed; tiles 1 .. D-1 may still be in flight
  const int my_pieces = (wave + WAVES * (PPW - 1) < NPIECE) ? PPW : PPW - 1;   // wave-uniform
  auto wait_keep = [&](int groups) {   // wait until at most `groups` tiles of this wave's pieces are outstanding
    const int n = groups * my_pieces;
    if (n <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n <= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  wait_keep(min(D, nt) - 1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr ((FLAGS & (256 | 512 | 1024)) != 0) {
    // de-synchronise the workgroups that share a CU: the one in the upper wave slots starts part of a tile period late
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    const bool late = (FLAGS & 1024) ? ((blockIdx.x + blockIdx.y) & 1) : (((hwid & 0xF) >> (WAVES == 8 ? 1 : 0)) & 1);
    if (late) {
      if constexpr ((FLAGS & 256) != 0) __builtin_amdgcn_s_sleep(6);     // ~384 cycles
      if constexpr ((FLAGS & 512) != 0) __builtin_amdgcn_s_sleep(12);    // ~768 cycles
    }
  }
  int rb = 0, wb = D % NBUF;   // ring slots of tile t and of tile t + D

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    const unsigned bo = (unsigned)(rb * BUF2);
    if (t + D < nt) stage(wb, t0 + 64 * D);
    // ---- S^T = K Q^T ----
    f32x16 s[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if (!NOQK || t == 0) {
      const unsigned ka = kaddr0 + bo;
      u32x4 a00 = lds_read128<0 * 3584 + 0>(ka), a01 = lds_read128<0 * 3584 + 32>(ka), a02 = lds_read128<0 * 3584 + 64>(ka);
      u32x4 a10 = lds_read128<1 * 3584 + 0>(ka), a11 = lds_read128<1 * 3584 + 32>(ka), a12 = lds_read128<1 * 3584 + 64>(ka);
      if constexpr ((FLAGS & 64) != 0) __builtin_amdgcn_s_setprio(1);
      wait_lgkm<5>(a00); mma(a00, qf[0], s[0]);
      wait_lgkm<4>(a01); mma(a01, qf[1], s[0]);
      wait_lgkm<3>(a02); mma(a02, qf[2], s[0]);
      wait_lgkm<2>(a10); mma(a10, qf[0], s[1]);
      wait_lgkm<1>(a11); mma(a11, qf[1], s[1]);
      wait_lgkm<0>(a12); mma(a12, qf[2], s[1]);
      if constexpr ((FLAGS & 64) != 0) __builtin_amdgcn_s_setprio(0);
    }
    if (t0 + 64 > p.Nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + ((r & 3) + 8 * (r >> 2) + 4 * h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    if (!NOMAX || t == 0) {
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
      const bool move = (t == 0) || mx > 24.0f;
      if (__builtin_amdgcn_ballot_w64(move) != 0) {
        const float m_new = move ? bf16_ceil(m_run + mx) : m_run;
        const float delta = m_new - m_run;
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
        if (h == MREF_HALF) {
          V16 v;
          v.u = qf[MREF_STEP];
          v.e[MREF_ELEM] = (bf16)(-m_new);
          qf[MREF_STEP] = v.u;
        }
      }
    }
    if (!NOEXP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    }
    // ---- O^T += V^T P^T ----
    if (!NOPV || t == 0) {
      const unsigned va0 = vaddr0 + bo + vhalf[0], va1 = vaddr0 + bo + vhalf[1];
#define AF_PV_STEP(KB, S2)                                                                                           \
      {                                                                                                              \
        constexpr int RO = ((KB) * 32 + (S2) * 16) * VROW2;                                                          \
        u32x2 lo0 = lds_read_tr64<RO>(va0), hi0 = lds_read_tr64<RO + 8 * VROW2>(va0);                                \
        u32x2 lo1 = lds_read_tr64<RO>(va1), hi1 = lds_read_tr64<RO + 8 * VROW2>(va1);                                \
        V16 pb;                                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) pb.e[j] = (bf16)s[KB][8 * (S2) + j];                           \
        wait_lgkm2<2>(lo0, hi0);                                                                                     \
        mma(u32x4{lo0.x, lo0.y, hi0.x, hi0.y}, pb.u, o[0]);                                                     \
        wait_lgkm2<0>(lo1, hi1);                                                                                     \
        mma(u32x4{lo1.x, lo1.y, hi1.x, hi1.y}, pb.u, o[1]);                                                     \
      }
      if constexpr ((FLAGS & 64) != 0) __builtin_amdgcn_s_setprio(1);
      AF_PV_STEP(0, 0) AF_PV_STEP(0, 1) AF_PV_STEP(1, 0) AF_PV_STEP(1, 1)
      if constexpr ((FLAGS & 64) != 0) __builtin_amdgcn_s_setprio(0);
#undef AF_PV_STEP
    } else {
      asm volatile("" ::"v"(s[0][0]), "v"(s[0][5]), "v"(s[1][3]), "v"(s[1][15]));
#pragma unroll
      for (int r = 0; r < 16; ++r) o[0][r] += s[0][r] + s[1][r];
    }
    wait_keep(min(D - 1, nt - 2 - t));                 // own pieces of tile t+1 landed (tiles t+2 .. stay in flight)
    __builtin_amdgcn_s_barrier();                      // everyone's pieces landed; everyone is done reading tile t
    rb = rb + 1 == NBUF ? 0 : rb + 1;
    wb = wb + 1 == NBUF ? 0 : wb + 1;
  }
  constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
  const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int dd = 32 * d + 8 * g + 4 * h;
        if (dd < DH) {
          Q4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = (bf16)(o[d][4 * g + e] * inv);
          *reinterpret_cast<uint2*>(O + (long)q * p.ldo + dd) = ov.u;
        }
      }
  }
}
template <int FLAGS, int WAVES, int NBUF> static void launch_ring(const P& p, int B, hipStream_t s) {
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)&attn_ring<FLAGS, WAVES, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * BUF2)); set = true; }
  dim3 grid((p.Nq + WAVES * 32 - 1) / (WAVES * 32), p.H, B);
  hipLaunchKernelGGL((attn_ring<FLAGS, WAVES, NBUF>), grid, dim3(WAVES * 64), NBUF * BUF2, s, p);
}
template <int FLAGS> static void launch_dma(const P& p, int B, hipStream_t s) {
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)&attn_dma<FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF2)); set = true; }
  dim3 grid((p.Nq + 127) / 128, p.H, B);
  hipLaunchKernelGGL((attn_dma<FLAGS>), grid, dim3(256), 2 * BUF2, s, p);
}

// ---------------------------------------------------------------------------------------------------------------
// Ping-pong variant: 8 waves = two groups (waves 0-3 / 4-7 = SIMD partners) that run the SAME program one barrier
// apart, two segments per tile:
//     Seg V(t): softmax of tile t (32 v_exp + 16 v_cvt_pk: VALU / transcendental pipe), LDS-DMA issue, counted wait
//     Seg M(t): O^T += V^T P^T of tile t, then S^T = K Q^T of tile t+1 (14 MFMAs + their LDS fragment reads)
// so that on every SIMD one wave of a pair multiplies while its partner exponentiates.  Barrier i opens interval i;
// group 0 runs Seg V(t) in interval 2t and Seg M(t) in 2t+1, group 1 one interval later.
// Ring of NBUF tiles: group g stages its pieces of tile u in its Seg V(u - Lg) and waits for them at the end of its
// Seg V(u - Wg) with L0 = NBUF-2, W0 = 1, L1 = NBUF-1, W1 = 2 (first reader of tile u: K by group 0 in interval 2u-1;
// last reader of tile u: V by group 1 in interval 2u+2) -> NBUF-3 tiles of flight time for every piece.
// ---------------------------------------------------------------------------------------------------------------
template <int FLAGS, int NBUF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu((FLAGS & 128) ? 2 : 4, (FLAGS & 128) ? 2 : 4))) void attn_pp(const P p) {
  constexpr bool NOEXP = FLAGS & 4, NOPV = FLAGS & 8, NOQK = FLAGS & 16, PRIO = FLAGS & 64;
  constexpr int WAVES = 8, NT = 512, PPW = 2;
  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = wave >> 2;
  const int h = lane >> 5, l31 = lane & 31;
  const int head = blockIdx.y, b = blockIdx.z;
  const int q = blockIdx.x * 256 + wave * 32 + l31;
  const bool q_ok = q < p.Nq;
  const bf16* Q = p.q + (long)b * p.bsq + head * p.hsq;
  const bf16* K = p.k + (long)b * p.bsk + head * p.hsk;
  const bf16* V = p.v + (long)b * p.bsv + head * p.hsv;
  bf16* O = p.o + (long)b * p.bso + head * DH;
  const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(K), 0, (p.Nk - 1) * p.ldk * 2 + DH * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(V), 0, (p.Nk - 1) * p.ldv * 2 + DH * 2, 0x00020000);
  const int nt = (p.Nk + 63) / 64;
  const int Lg = g == 0 ? NBUF - 2 : NBUF - 1, Wg = g == 0 ? 1 : 2;

  // Q fragments first (oldest VMEM operations of the wave)
  const float sl2 = p.scale * 1.44269504088896340736f;
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / 2;
    V16 v;
    v.u = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v.u = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
#pragma unroll
    for (int e = 0; e < 8; ++e) v.e[e] = (bf16)((float)v.e[e] * sl2);
    qf[s] = v.u;
  }
  // ---- DMA pieces of this wave: piece = wave + 8 j (j < 2); pieces 0..6 = K image, 7..14 = V image ----
  unsigned voff[PPW];
  bool act[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int piece = wave + WAVES * j;
    if (piece < KP) {
      const int gg = piece * 64 + lane, row = gg / 7, c = gg - row * 7;
      act[j] = c < 5;
      voff[j] = (unsigned)(row * p.ldk * 2 + c * 16);
    } else {
      const int gg = (piece - KP) * 64 + lane, row = gg >> 3, c = (gg & 7) ^ (((row >> 1) & 1) << 2);
      act[j] = c < 5 && piece < NPIECE;
      voff[j] = (unsigned)(row * p.ldv * 2 + c * 16);
    }
  }
  const int my_pieces = (wave + WAVES < NPIECE) ? 2 : 1;   // wave-uniform
  auto stage = [&](int u) {      // this wave's pieces of tile u into slot u % NBUF
    char* base = smem + (u % NBUF) * BUF2;
    const int t0 = u * 64;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const int piece = wave + WAVES * j;
      if (piece >= NPIECE) continue;
      if (piece < KP) { if (act[j]) lds_dma16(rs_k, base + piece * 1024, voff[j], (unsigned)(t0 * p.ldk * 2)); }
      else { if (act[j]) lds_dma16(rs_v, base + K2_BYTES + (piece - KP) * 1024, voff[j], (unsigned)(t0 * p.ldv * 2)); }
    }
  };
  auto wait_keep = [&](int groups) {   // at most `groups` staged tiles of this wave may still be in flight
    const int n = groups <= 0 ? 0 : groups * my_pieces;
    if (n == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (n == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (n == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if (n == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if (n == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if (n <= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  };
  // prologue: tiles 0 .. Lg-1
  for (int u = 0; u < Lg && u < nt; ++u) stage(u);
  for (int idx = tid; idx < NBUF * 320; idx += NT) {
    const int bufi = idx / 320, r = idx - bufi * 320;
    char* base = smem + bufi * BUF2;
    if (r < 128) {
      const int row = r >> 1, c = 5 + (r & 1);
      *reinterpret_cast<uint4*>(base + row * KROW2 + c * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    } else {
      const int rr2 = r - 128, row = rr2 / 3, c = 5 + (rr2 - row * 3);
      const int pos = c ^ (((row >> 1) & 1) << 2);
      *reinterpret_cast<uint4*>(base + K2_BYTES + row * VROW2 + pos * 16) = make_uint4(c == 5 ? 0x3F80u : 0u, 0, 0, 0);
    }
  }
  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = 0.f;
  const unsigned lds0 = (unsigned)(__UINTPTR_TYPE__)((__attribute__((address_space(3))) char*)smem);
  const unsigned kaddr0 = lds0 + (unsigned)(l31 * KROW2 + 16 * h);
  const int tq = (lane & 15) >> 2, tp = lane & 3, gi = (lane >> 4) & 1;
  const int vrow0 = 4 * h + tq, sw = (vrow0 >> 1) & 1;
  const unsigned vaddr0 = lds0 + (unsigned)(K2_BYTES + vrow0 * VROW2 + 32 * gi + 8 * tp);
  const unsigned vh0 = (unsigned)(64 * (0 ^ sw)), vh1 = (unsigned)(64 * (1 ^ sw));

  f32x16 s[2];
  auto qk = [&](int u) {     // S^T = K Q^T of tile u
    const unsigned ka = kaddr0 + (unsigned)((u % NBUF) * BUF2);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
    if (NOQK && u > 0) return;
    u32x4 a00 = lds_read128<0 * 3584 + 0>(ka), a01 = lds_read128<0 * 3584 + 32>(ka), a02 = lds_read128<0 * 3584 + 64>(ka);
    u32x4 a10 = lds_read128<1 * 3584 + 0>(ka), a11 = lds_read128<1 * 3584 + 32>(ka), a12 = lds_read128<1 * 3584 + 64>(ka);
    wait_lgkm<5>(a00); mma(a00, qf[0], s[0]);
    wait_lgkm<4>(a01); mma(a01, qf[1], s[0]);
    wait_lgkm<3>(a02); mma(a02, qf[2], s[0]);
    wait_lgkm<2>(a10); mma(a10, qf[0], s[1]);
    wait_lgkm<1>(a11); mma(a11, qf[1], s[1]);
    wait_lgkm<0>(a12); mma(a12, qf[2], s[1]);
  };
  // wait for tile 0 (group 1: tiles 0 and 1, whose first reader runs before group 1's first Seg V ends)
  wait_keep(min(Lg, nt) - Wg);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();                 // barrier -1
  if (g == 1) __builtin_amdgcn_s_barrier();     // group 1 idles through interval -1
  qk(0);                                        // Seg M(-1)
  __builtin_amdgcn_s_barrier();

  for (int t = 0; t < nt; ++t) {
    const int t0 = t * 64;
    // ---------------- Seg V(t) ----------------
    if (t0 + 64 > p.Nk) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (t0 + 32 * kb + ((r & 3) + 8 * (r >> 2) + 4 * h) >= p.Nk) s[kb][r] = -INFINITY;
    }
    if (t == 0) {
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
      const float m_new = bf16_ceil(mx);
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] -= m_new;
      m_run = m_new;
      if (h == MREF_HALF) {
        V16 v;
        v.u = qf[MREF_STEP];
        v.e[MREF_ELEM] = (bf16)(-m_new);
        qf[MREF_STEP] = v.u;
      }
    }
    if (!NOEXP) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]);
    }
    V16 pb[2][2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) pb[kb][s2].e[j] = (bf16)s[kb][8 * s2 + j];
    if (t + Lg < nt) stage(t + Lg);
    wait_keep(min(t + Lg, nt - 1) - (t + Wg));
    __builtin_amdgcn_sched_barrier(0);   // keep the segments apart: hipcc moves register-only work across s_barrier
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---------------- Seg M(t) ----------------
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
    {
      // All 16 transposed V reads of the tile go out first (they reuse the registers of s, dead until QK(t+1)); the six K
      // reads of tile t+1 follow once half of the V fragments have been consumed.  LDS returns in order: lgkmcnt(N) = the
      // N youngest reads may still be outstanding.
      const unsigned bo = (unsigned)((t % NBUF) * BUF2);
      const unsigned va0 = vaddr0 + bo + vh0, va1 = vaddr0 + bo + vh1;
      const unsigned ka = kaddr0 + (unsigned)(((t + 1) % NBUF) * BUF2);
      const bool has_next = t + 1 < nt;
      u32x2 vf[4][4];
#define AF_VREAD(ST, KB, S2)                                                               \
      vf[ST][0] = lds_read_tr64<((KB) * 32 + (S2) * 16) * VROW2>(va0);                        \
      vf[ST][1] = lds_read_tr64<((KB) * 32 + (S2) * 16 + 8) * VROW2>(va0);                    \
      vf[ST][2] = lds_read_tr64<((KB) * 32 + (S2) * 16) * VROW2>(va1);                        \
      vf[ST][3] = lds_read_tr64<((KB) * 32 + (S2) * 16 + 8) * VROW2>(va1);
      AF_VREAD(0, 0, 0) AF_VREAD(1, 0, 1) AF_VREAD(2, 1, 0) AF_VREAD(3, 1, 1)
#undef AF_VREAD
#define AF_PVM(ST, KB, S2, N0, N1)                                                                      \
      wait_lgkm2<N0>(vf[ST][0], vf[ST][1]);                                                              \
      mma(u32x4{vf[ST][0].x, vf[ST][0].y, vf[ST][1].x, vf[ST][1].y}, pb[KB][S2].u, o[0]);               \
      wait_lgkm2<N1>(vf[ST][2], vf[ST][3]);                                                              \
      mma(u32x4{vf[ST][2].x, vf[ST][2].y, vf[ST][3].x, vf[ST][3].y}, pb[KB][S2].u, o[1]);
      if (has_next) {
        AF_PVM(0, 0, 0, 14, 12) AF_PVM(1, 0, 1, 10, 8) AF_PVM(2, 1, 0, 6, 4)
        u32x4 a00 = lds_read128<0 * 3584 + 0>(ka), a01 = lds_read128<0 * 3584 + 32>(ka), a02 = lds_read128<0 * 3584 + 64>(ka);
        AF_PVM(3, 1, 1, 5, 3)
        u32x4 a10 = lds_read128<1 * 3584 + 0>(ka), a11 = lds_read128<1 * 3584 + 32>(ka), a12 = lds_read128<1 * 3584 + 64>(ka);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) s[kb][r] = 0.f;
        wait_lgkm<5>(a00); mma(a00, qf[0], s[0]);
        wait_lgkm<4>(a01); mma(a01, qf[1], s[0]);
        wait_lgkm<3>(a02); mma(a02, qf[2], s[0]);
        wait_lgkm<2>(a10); mma(a10, qf[0], s[1]);
        wait_lgkm<1>(a11); mma(a11, qf[1], s[1]);
        wait_lgkm<0>(a12); mma(a12, qf[2], s[1]);
      } else {
        AF_PVM(0, 0, 0, 14, 12) AF_PVM(1, 0, 1, 10, 8) AF_PVM(2, 1, 0, 6, 4) AF_PVM(3, 1, 1, 2, 0)
      }
#undef AF_PVM
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (g == 0) __builtin_amdgcn_s_barrier();     // both groups execute the same number of barriers
  constexpr int rr = DH % 32, ob = DH / 32, oreg = (rr & 3) + 4 * (rr >> 3), oh = (rr >> 2) & 1;
  const float l_tot = __shfl(o[ob][oreg], l31 + 32 * oh, 64);
  const float inv = 1.0f / l_tot;
  if (q_ok) {
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const int dd = 32 * d + 8 * gq + 4 * h;
        if (dd < DH) {
          Q4 ov;
#pragma unroll
          for (int e = 0; e < 4; ++e) ov.e[e] = (bf16)(o[d][4 * gq + e] * inv);
          *reinterpret_cast<uint2*>(O + (long)q * p.ldo + dd) = ov.u;
        }
      }
  }
}
template <int FLAGS, int NBUF> static void launch_pp(const P& p, int B, hipStream_t s) {
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)&attn_pp<FLAGS, NBUF>, hipFuncAttributeMaxDynamicSharedMemorySize, NBUF * BUF2)); set = true; }
  dim3 grid((p.Nq + 255) / 256, p.H, B);
  hipLaunchKernelGGL((attn_pp<FLAGS, NBUF>), grid, dim3(512), NBUF * BUF2, s, p);
}

// naive fp32 reference for one (batch, head): one thread per query
__global__ void attn_ref(const P p, int b, int head, float* out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= p.Nq) return;
  const bf16* Q = p.q + (long)b * p.bsq + head * p.hsq + (long)q * p.ldq;
  const bf16* K = p.k + (long)b * p.bsk + head * p.hsk;
  const bf16* V = p.v + (long)b * p.bsv + head * p.hsv;
  float m = -INFINITY, l = 0.f, acc[DH];
  for (int d = 0; d < DH; ++d) acc[d] = 0.f;
  for (int j = 0; j < p.Nk; ++j) {
    float s = 0.f;
    for (int d = 0; d < DH; ++d) s += (float)Q[d] * (float)K[(long)j * p.ldk + d];
    s *= p.scale;
    const float mn = fmaxf(m, s), a = expf(m - mn), pe = expf(s - mn);
    l = l * a + pe;
    for (int d = 0; d < DH; ++d) acc[d] = acc[d] * a + pe * (float)V[(long)j * p.ldv + d];
    m = mn;
  }
  for (int d = 0; d < DH; ++d) out[(long)q * DH + d] = acc[d] / l;
}

template <int FLAGS, int WAVES> static void launch(const P& p, int B, hipStream_t s) {
  static bool set = false;
  if (!set) { CK(hipFuncSetAttribute((const void*)&attn_lab<FLAGS, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * TILE)); set = true; }
  dim3 grid((p.Nq + WAVES * 32 - 1) / (WAVES * 32), p.H, B);
  hipLaunchKernelGGL((attn_lab<FLAGS, WAVES>), grid, dim3(WAVES * 64), 2 * TILE, s, p);
}

struct Variant { const char* name; void (*fn)(const P&, int, hipStream_t); bool valid; bool headmajor = false; };

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  const int B = 16, N = 4096, H = 8, C = H * DH;
  const size_t nqkv = (size_t)B * N * 3 * C, no = (size_t)B * N * C;
  std::vector<bf16> hq(nqkv);
  srand(1);
  for (size_t i = 0; i < nqkv; ++i) {
    float u = 0.f;
    for (int j = 0; j < 4; ++j) u += (float)rand() / RAND_MAX;
    hq[i] = (bf16)((u - 2.0f) * 1.7f);   // ~N(0,1)
  }
  bf16 *dqkv, *dout, *dout0;
  float* dref;
  CK(hipMalloc(&dqkv, nqkv * 2)); CK(hipMalloc(&dout, no * 2)); CK(hipMalloc(&dout0, no * 2)); CK(hipMalloc(&dref, (size_t)N * DH * 4));
  CK(hipMemcpy(dqkv, hq.data(), nqkv * 2, hipMemcpyHostToDevice));
  P p;
  p.q = dqkv; p.k = dqkv + C; p.v = dqkv + 2 * C; p.o = dout;
  p.ldq = p.ldk = p.ldv = 3 * C; p.ldo = C;
  p.bsq = p.bsk = p.bsv = (long)N * 3 * C; p.bso = (long)N * C;
  p.Nq = N; p.Nk = N; p.H = H; p.scale = 1.0f / sqrtf((float)DH);
  p.hsq = p.hsk = p.hsv = DH;
  // head-major copy: [which][head][b * N + n][dh]
  std::vector<bf16> hm(nqkv);
  for (int w = 0; w < 3; ++w)
    for (int hh = 0; hh < H; ++hh)
      for (long m = 0; m < (long)B * N; ++m)
        memcpy(&hm[(((size_t)w * H + hh) * B * N + m) * DH], &hq[(size_t)m * 3 * C + w * C + hh * DH], DH * 2);
  bf16* dhm;
  CK(hipMalloc(&dhm, nqkv * 2));
  CK(hipMemcpy(dhm, hm.data(), nqkv * 2, hipMemcpyHostToDevice));
  P ph = p;
  ph.q = dhm; ph.k = dhm + (size_t)H * B * N * DH; ph.v = dhm + (size_t)2 * H * B * N * DH;
  ph.ldq = ph.ldk = ph.ldv = DH;
  ph.bsq = ph.bsk = ph.bsv = (long)N * DH;
  ph.hsq = ph.hsk = ph.hsv = (long)B * N * DH;

  std::vector<Variant> vs = {
      {"V0 shipped (4 waves, running reference)", launch<0, 4>, true},
      {"V1 nomax", launch<1, 4>, true},
      {"V2 xcd", launch<2, 4>, true},
      {"V3 nomax+xcd", launch<3, 4>, true},
      {"V4 8 waves", launch<0, 8>, true},
      {"V5 8 waves nomax+xcd", launch<3, 8>, true},
      {"D0 dma staging, running reference", launch_dma<0>, true},
      {"D1 dma staging, nomax", launch_dma<1>, true},
      {"R1 ring 4 waves x 2 buffers (= D1)", launch_ring<1, 4, 2>, true},
      {"R2 ring 4 waves x 3 buffers", launch_ring<1, 4, 3>, true},
      {"R3 ring 8 waves x 2 buffers", launch_ring<1, 8, 2>, true},
      {"R4 ring 8 waves x 3 buffers", launch_ring<1, 8, 3>, true},
      {"R5 ring 8 waves x 4 buffers", launch_ring<1, 8, 4>, true},
      {"R6 ring 8 waves x 5 buffers", launch_ring<1, 8, 5>, true},
      {"R4m ring 8 waves x 3 buffers, running max", launch_ring<0, 8, 3>, true},
      {"P4 ping-pong 4 buffers head-major", launch_pp<1, 4>, true, true},
      {"P5 ping-pong 5 buffers head-major", launch_pp<1, 5>, true, true},
      {"P5p ping-pong 5 buffers head-major, prio in Seg M", launch_pp<1 | 64, 5>, true, true},
      {"P5i ping-pong 5 buffers interleaved layout", launch_pp<1, 5>, true, false},
      {"Q5 ping-pong 5 buffers, 1 WG/CU (256 VGPR)", launch_pp<1 | 128, 5>, true, true},
      {"Q8 ping-pong 8 buffers, 1 WG/CU", launch_pp<1 | 128, 8>, true, true},
      {"Q8p ping-pong 8 buffers, 1 WG/CU, prio", launch_pp<1 | 128 | 64, 8>, true, true},
      {"Q8i ping-pong 8 buffers, 1 WG/CU, interleaved layout", launch_pp<1 | 128, 8>, true, false},
      {"QA1 ping-pong 8 1WG no exp", launch_pp<1 | 128 | 4, 8>, false, true},
      {"PA1 ping-pong 5 no exp", launch_pp<1 | 4, 5>, false, true},
      {"H3 ring 8x2 head-major", launch_ring<1, 8, 2>, true, true},
      {"H4 ring 8x3 head-major", launch_ring<1, 8, 3>, true, true},
      {"H5 ring 8x4 head-major", launch_ring<1, 8, 4>, true, true},
      {"H2 ring 4x2 head-major", launch_ring<1, 4, 2>, true, true},
      {"H2b ring 4x3 head-major", launch_ring<1, 4, 3>, true, true},
      {"H4x ring 8x3 head-major xcd", launch_ring<3, 8, 3>, true, true},
      {"H4m ring 8x3 head-major running max", launch_ring<0, 8, 3>, true, true},
      {"HA5 ring 8x3 head-major no exp/PV/QK", launch_ring<1 | 4 | 8 | 16, 8, 3>, false, true},
      {"HA1 ring 8x3 head-major no exp", launch_ring<1 | 4, 8, 3>, false, true},
      {"HA2 ring 8x3 head-major no PV", launch_ring<1 | 8, 8, 3>, false, true},
      {"HA3 ring 8x3 head-major no QK", launch_ring<1 | 16, 8, 3>, false, true},
      {"HV0 shipped kernel on head-major", launch<0, 4>, true, true},
      {"R3p ring 8x2 setprio around MFMA", launch_ring<1 | 64, 8, 2>, true},
      {"S1 ring 8x2 stagger 384 by wave slot", launch_ring<1 | 256, 8, 2>, true},
      {"S2 ring 8x2 stagger 768 by wave slot", launch_ring<1 | 512, 8, 2>, true},
      {"S3 ring 8x2 stagger 1152 by wave slot", launch_ring<1 | 256 | 512, 8, 2>, true},
      {"S4 ring 8x2 stagger 384 by block parity", launch_ring<1 | 256 | 1024, 8, 2>, true},
      {"S5 ring 4x2 stagger 384 by wave slot", launch_ring<1 | 256, 4, 2>, true},
      {"S6 ring 4x2 stagger 768 by wave slot", launch_ring<1 | 512, 4, 2>, true},
      {"R3x ring 8x2 xcd", launch_ring<3, 8, 2>, true},
      {"R4x ring 8x3 xcd", launch_ring<3, 8, 3>, true},
      {"R5x ring 8x4 xcd", launch_ring<3, 8, 4>, true},
      {"R2x ring 4x3 xcd", launch_ring<3, 4, 3>, true},
      {"RA5x ring 8x4 xcd no exp/PV/QK", launch_ring<3 | 4 | 8 | 16, 8, 4>, false},
      {"RA5 ring 8x4 no exp/PV/QK", launch_ring<1 | 4 | 8 | 16, 8, 4>, false},
      {"RA1 ring 8x4 no exp", launch_ring<1 | 4, 8, 4>, false},
      {"RA2 ring 8x4 no PV", launch_ring<1 | 8, 8, 4>, false},
      {"RA3 ring 8x4 no QK", launch_ring<1 | 16, 8, 4>, false},
      {"DA1 dma nomax no exp", launch_dma<1 | 4>, false},
      {"DA2 dma nomax no PV", launch_dma<1 | 8>, false},
      {"DA3 dma nomax no QK", launch_dma<1 | 16>, false},
      {"DA5 dma nomax no exp/PV/QK", launch_dma<1 | 4 | 8 | 16>, false},
      {"A1 nomax, no exp", launch<1 | 4, 4>, false},
      {"A2 nomax, no PV", launch<1 | 8, 4>, false},
      {"A3 nomax, no QK", launch<1 | 16, 4>, false},
      {"A4 nomax, no staging", launch<1 | 32, 4>, false},
      {"A5 nomax, no exp no PV no QK (barriers+staging only)", launch<1 | 4 | 8 | 16, 4>, false},
      {"A6 nomax, no exp no staging", launch<1 | 4 | 32, 4>, false},
  };
  hipStream_t st;
  CK(hipStreamCreate(&st));
  // reference for (b = 3, head = 5)
  hipLaunchKernelGGL(attn_ref, dim3(N / 64), dim3(64), 0, st, p, 3, 5, dref);
  std::vector<float> href((size_t)N * DH);
  CK(hipStreamSynchronize(st));
  CK(hipMemcpy(href.data(), dref, href.size() * 4, hipMemcpyDeviceToHost));
  std::vector<bf16> h0(no), h1(no);
  std::vector<std::vector<float>> times(vs.size());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (size_t v = 0; v < vs.size(); ++v) {   // warm + check
    CK(hipMemsetAsync(dout, 0, no * 2, st));
    vs[v].fn(vs[v].headmajor ? ph : p, B, st);
    CK(hipStreamSynchronize(st));
    CK(hipGetLastError());
    if (vs[v].valid) {
      CK(hipMemcpy(h1.data(), dout, no * 2, hipMemcpyDeviceToHost));
      double eref = 0, scale = 0, e0d = 0;
      for (int q = 0; q < N; ++q)
        for (int d = 0; d < DH; ++d) {
          const float got = (float)h1[((size_t)3 * N + q) * C + 5 * DH + d], ref = href[(size_t)q * DH + d];
          eref = std::max(eref, (double)fabsf(got - ref));
          scale = std::max(scale, (double)fabsf(ref));
        }
      if (v == 0) h0 = h1;
      else for (size_t i = 0; i < no; ++i) e0d = std::max(e0d, (double)fabsf((float)h1[i] - (float)h0[i]));
      printf("check %-48s max|got-ref| = %.3e (scale %.3f)  max|got-V0| = %.3e\n", vs[v].name, eref, scale, e0d);
    }
  }
  for (int r = 0; r < reps; ++r)
    for (size_t v = 0; v < vs.size(); ++v) {
      CK(hipEventRecord(e0, st));
      vs[v].fn(vs[v].headmajor ? ph : p, B, st);
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      times[v].push_back(ms * 1e3f);
    }
  const double flop = 4.0 * B * H * (double)N * N * DH;
  for (size_t v = 0; v < vs.size(); ++v) {
    std::sort(times[v].begin(), times[v].end());
    const float med = times[v][times[v].size() / 2], mn = times[v][0];
    printf("%-56s median %8.1f us  min %8.1f us  %7.1f TF/s (alg, median)\n", vs[v].name, med, mn, flop / (med * 1e-6) / 1e12);
  }
  return 0;
}
