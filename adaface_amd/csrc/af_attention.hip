// Flash-style attention for gfx950: softmax(Q K^T * scale) V without materialising
// the [N, S] score matrix.  Serves both uses of the reference's CrossAttention
// class (ldm/modules/attention.py:172-257): attn1 = self-attention over
// N in {4096,1024,256,64} image tokens and attn2 = cross-attention over the 77
// context tokens; 8 heads, dh = C/8 in {40,80,160}.  The reference computes
// einsum -> softmax -> einsum with a full `sim` tensor (attention.py:199,238,240).
//
// Structure (wave64, MFMA 32x32):
//   * grid (ceil(Nq/128), heads, batch); 4 waves, each owns 32 query rows.
//   * K/V are staged per 64-key tile in LDS: K row-major (row stride padded by
//     16 B -> conflict-free ds_read_b128), V TRANSPOSED ([d][key]) so the PV
//     product reads its A operand with 8/16-byte LDS reads.
//   * swapped QK^T: S^T = K . Q^T, so a lane holds ONE query column and 16 keys
//     per 32x32 block in registers; the online-softmax max/sum are register
//     reductions plus one cross-half shuffle.
//   * the S^T accumulator is fed straight back as the B operand of O^T = V^T . P^T
//     (accumulator-as-operand: register r of lane-half h is key (r&3)+8(r>>2)+4h,
//     the V^T fragment is read in that same key order), no LDS round trip for P.
//   * dh is padded in LDS only: QK^T K-dim to a multiple of 32 bytes, O to
//     32-row blocks; HBM traffic is exactly the unpadded Q, K, V, O.
#include "af_common.h"
#include <math.h>

template <typename T, int DH> struct AttnCfg {
  static constexpr int EPC = 16 / sizeof(T);
  static constexpr int FS = (DH * (int)sizeof(T) + 31) / 32;   // 32-byte steps along d
  static constexpr int KROW = FS * 32 + 16;                    // bytes
  static constexpr int DB = (DH + 31) / 32;                    // 32-wide d blocks
  static constexpr int VROW = 64 * (int)sizeof(T) + (sizeof(T) == 2 ? 8 : 16);
  static constexpr int K_BYTES = 64 * KROW;
  static constexpr int V_BYTES = DB * 32 * VROW;
  static constexpr int LDS_BYTES = K_BYTES + V_BYTES;
};

template <typename T, int DH>
__global__ __launch_bounds__(256) void attn_kernel(const AttnParams p) {
  using C = AttnCfg<T, DH>;
  constexpr int EPC = C::EPC, FS = C::FS, KROW = C::KROW, DB = C::DB, VROW = C::VROW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ks = smem;
  char* vt = smem + C::K_BYTES;

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

  // Q fragments (B operand of S^T = K Q^T): lane (q, h) holds d = (32s+16h)/sizeof(T) ...
  uint4 qf[FS];
#pragma unroll
  for (int s = 0; s < FS; ++s) {
    const int d0 = (32 * s + 16 * h) / (int)sizeof(T);
    uint4 v = make_uint4(0, 0, 0, 0);
    if (q_ok && d0 < DH) v = *reinterpret_cast<const uint4*>(Q + (long)q * p.ldq + d0);
    qf[s] = v;
  }

  f32x16 o[DB];
#pragma unroll
  for (int d = 0; d < DB; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float sl2 = p.scale * 1.44269504088896340736f;

  for (int t0 = 0; t0 < p.Nk; t0 += 64) {
    __syncthreads();
    // ---- stage K tile: [64][KROW], chunks beyond DH and keys beyond Nk are zero ----
    {
      constexpr int CPR = FS * 2;  // 16-byte chunks per LDS row
      for (int idx = tid; idx < 64 * CPR; idx += 256) {
        const int row = idx / CPR, ch = idx - row * CPR;
        const int key = t0 + row;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (key < p.Nk && ch * EPC < DH) v = *reinterpret_cast<const uint4*>(K + (long)key * p.ldk + ch * EPC);
        *reinterpret_cast<uint4*>(ks + row * KROW + ch * 16) = v;
      }
    }
    // ---- stage V tile transposed: vt[d][key]; rows d >= DH zero, keys >= Nk zero ----
    {
      constexpr int NCH = DB * 32 / EPC;  // chunks per key incl. padding rows
      const int krow = tid & 63;
      const int key = t0 + krow;
      for (int ch = tid >> 6; ch < NCH; ch += 4) {
        Vec16<T> v;
        v.u = make_uint4(0, 0, 0, 0);
        if (key < p.Nk && ch * EPC < DH) v.u = *reinterpret_cast<const uint4*>(V + (long)key * p.ldv + ch * EPC);
#pragma unroll
        for (int e = 0; e < EPC; ++e)
          *reinterpret_cast<T*>(vt + (ch * EPC + e) * VROW + krow * (int)sizeof(T)) = v.e[e];
      }
    }
    __syncthreads();

    // ---- S^T = K Q^T : two 32-key blocks ----
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
    // ---- online softmax (per lane = per query; halves hold disjoint keys) ----
    float mx = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = t0 + 32 * kb + acc_row(r, h);
        float v = s[kb][r] * sl2;
        if (key >= p.Nk) v = -INFINITY;
        s[kb][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = exp2f(m_run - m_new);
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float pv = exp2f(s[kb][r] - m_new);
        s[kb][r] = pv;
        psum += pv;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
#pragma unroll
    for (int d = 0; d < DB; ++d)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[d][r] *= alpha;

    // ---- O^T += V^T P^T ----
    if constexpr (sizeof(T) == 2) {
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          Vec16<T> pb;
#pragma unroll
          for (int j = 0; j < 8; ++j) pb.e[j] = from_f32<T>(s[kb][8 * s2 + j]);
          const int key0 = 32 * kb + 16 * s2 + 4 * h;
#pragma unroll
          for (int d = 0; d < DB; ++d) {
            const char* row = vt + (32 * d + l31) * VROW;
            const uint2 lo = *reinterpret_cast<const uint2*>(row + key0 * 2);
            const uint2 hi = *reinterpret_cast<const uint2*>(row + (key0 + 8) * 2);
            const uint4 a = make_uint4(lo.x, lo.y, hi.x, hi.y);
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
            const uint4 a = *reinterpret_cast<const uint4*>(vt + (32 * d + l31) * VROW + key0 * 4);
            Mma<T>::step(a, pb.u, o[d]);
          }
        }
    }
  }

  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  const float inv = 1.0f / l_tot;
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

template <typename T, int DH> static int launch_attn(const AttnParams& p, int B, hipStream_t stream) {
  using C = AttnCfg<T, DH>;
  static bool attr_set = false;
  if (!attr_set) {
    HIP_CHECK_RET(hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_kernel<T, DH>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS_BYTES));
    attr_set = true;
  }
  dim3 grid((p.Nq + 127) / 128, p.H, B);
  hipLaunchKernelGGL((attn_kernel<T, DH>), grid, dim3(256), C::LDS_BYTES, stream, p);
  HIP_CHECK_RET(hipGetLastError());
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
template int af_launch_attention<float>(const AttnParams&, int, int, hipStream_t);
