// mhsa_x3.hip -- the attention core of the "bf16x3" parity mode (row B2 at north_star's 1e-4): softmax(Q K^T / 8 + pad mask) V from FP32
// q, k, v on the bf16 matrix pipe, flash style (no (B, heads, T, T) score tensor), with every matrix operand split in two bf16 terms
//     x = x_h + x_l (+ 2^-17 |x|),  x_h = bf16(x), x_l = bf16(x - x_h)
// and every product taken as  a_h b_h + a_l b_h + a_h b_l  (exact bf16 products, fp32 accumulation; the dropped a_l b_l term is 2^-18 relative):
//     S^T = K Q^T      3 x 8 v_mfma_f32_32x32x16_bf16 per 64-key tile (K_h Q_h, K_l Q_h, K_h Q_l)
//     P                exact online softmax in fp32 (row maximum, deferred rescale as in mhsa.hip's exact tile), exp2 with 1 / sqrt(64) log2(e) folded in
//     O^T += V^T P^T   P split like the other operands: V_h P_h, V_l P_h, V_h P_l
// Reference: the S3PRL BERT attention behind model.py:164 / runner.py:556-575 (scores + additive -10000 on padded keys -> softmax -> P V in fp32); a
// padded key's exp() is exactly 0 in fp32, so excluding keys >= lengths[b] is the same arithmetic.
// Layout / geometry as mhsa.hip (4 waves x 32 queries, 64-key tiles double buffered in LDS, kv_off swizzle, K as the A operand so that a lane holds
// 32 scores of ONE query row, V^T fragments through ds_read_b64_tr_b16); the tiles are split into their two bf16 terms WHILE they are staged
// (fp32 global -> registers -> {hi, lo} bf16 tiles in LDS: 4 x 8 KiB per buffer).  3 x the matrix work of the bf16 kernel; replaces the parity
// mode's materialised scores (two batched fp32 GEMMs + a softmax pass over 1.5 GB per layer at B = 32).
#include "common.h"
#include "bf16.h"
#include "mhsa_tile.h"

namespace se {

namespace {

struct Split8 {
  bf16x8 h, l;
};

__device__ __forceinline__ Split8 split8(const float4 a, const float4 b) {
  const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
  Split8 r;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const __bf16 h = (__bf16)v[j];
    r.h[j] = h;
    r.l[j] = (__bf16)(v[j] - (float)h);
  }
  return r;
}

constexpr int kXTile = kAK * kHD * 2;        // one bf16 tile: 8 KiB
constexpr int kXBuf = 4 * kXTile;            // K_h, K_l, V_h, V_l: 32 KiB per buffer

}  // namespace

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mhsa_x3_kernel(const float* __restrict__ qkv, const int32_t* __restrict__ lengths,
                                                                                                int T, int H, float* __restrict__ ctx,
                                                                                                uint16_t* __restrict__ ctx3, int Kp) {
  __shared__ __attribute__((aligned(16))) char smem[2 * kXBuf];      // 64 KiB

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hh = lane >> 5;
  // XCD-aware mapping of mhsa.hip: all query tiles of one (utterance, head) on one XCD (they re-read the same K / V through its L2)
  int b, head, qt;
  {
    const int nqt = gridDim.x, pairs = gridDim.y * gridDim.z;
    const int lin = blockIdx.x + nqt * (blockIdx.y + gridDim.y * blockIdx.z);
    if ((pairs & 7) == 0) {
      const int xcd = lin & 7, i = lin >> 3;
      const int pair = 8 * (i / nqt) + xcd;
      qt = i % nqt;
      head = pair % gridDim.y;
      b = pair / gridDim.y;
    } else {
      qt = blockIdx.x; head = blockIdx.y; b = blockIdx.z;
    }
  }
  const int q0 = qt * kAQ + wave * 32;
  const int ld = 3 * H;
  const int len = lengths ? min(max(lengths[b], 1), T) : T;
  const int nkt = (len + kAK - 1) / kAK;
  const float* base = qkv + (size_t)b * T * ld + head * kHD;

  // ---- Q fragments (B operand of S^T = K Q^T), both terms: lane -> query q0 + l31, d = 16 s + 8 hh .. + 7
  bf16x8 qh[4], ql[4];
  {
    const float* qp = base + (size_t)min(q0 + l31, T - 1) * ld + 8 * hh;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const Split8 t = split8(*reinterpret_cast<const float4*>(qp + 16 * s), *reinterpret_cast<const float4*>(qp + 16 * s + 4));
      qh[s] = t.h;
      ql[s] = t.l;
    }
  }

  // ---- staging: K and V tiles are 64 rows x 64 fp32; thread -> rows srow, srow + 32, the 8 columns of chunk sch (two float4 each)
  const int srow = tid >> 3, sch = tid & 7;
  const float* kp = base + H + sch * 8;
  const float* vp = base + 2 * H + sch * 8;
  float4 rk[4], rv[4];
  const int so0 = kv_off(srow, sch), so1 = kv_off(srow + 32, sch);
#define SE_X_ISSUE(kt)                                                                       \
  do {                                                                                       \
    const size_t r0 = (size_t)min((kt) * kAK + srow, T - 1) * ld;                            \
    const size_t r1 = (size_t)min((kt) * kAK + srow + 32, T - 1) * ld;                       \
    rk[0] = *reinterpret_cast<const float4*>(kp + r0);                                       \
    rk[1] = *reinterpret_cast<const float4*>(kp + r0 + 4);                                   \
    rk[2] = *reinterpret_cast<const float4*>(kp + r1);                                       \
    rk[3] = *reinterpret_cast<const float4*>(kp + r1 + 4);                                   \
    rv[0] = *reinterpret_cast<const float4*>(vp + r0);                                       \
    rv[1] = *reinterpret_cast<const float4*>(vp + r0 + 4);                                   \
    rv[2] = *reinterpret_cast<const float4*>(vp + r1);                                       \
    rv[3] = *reinterpret_cast<const float4*>(vp + r1 + 4);                                   \
  } while (0)
#define SE_X_WRITE(buf)                                                                      \
  do {                                                                                       \
    char* w_ = smem + (buf) * kXBuf;                                                         \
    const Split8 k0_ = split8(rk[0], rk[1]), k1_ = split8(rk[2], rk[3]);                     \
    const Split8 v0_ = split8(rv[0], rv[1]), v1_ = split8(rv[2], rv[3]);                     \
    *reinterpret_cast<bf16x8*>(w_ + so0) = k0_.h;                                            \
    *reinterpret_cast<bf16x8*>(w_ + so1) = k1_.h;                                            \
    *reinterpret_cast<bf16x8*>(w_ + kXTile + so0) = k0_.l;                                   \
    *reinterpret_cast<bf16x8*>(w_ + kXTile + so1) = k1_.l;                                   \
    *reinterpret_cast<bf16x8*>(w_ + 2 * kXTile + so0) = v0_.h;                               \
    *reinterpret_cast<bf16x8*>(w_ + 2 * kXTile + so1) = v1_.h;                               \
    *reinterpret_cast<bf16x8*>(w_ + 3 * kXTile + so0) = v0_.l;                               \
    *reinterpret_cast<bf16x8*>(w_ + 3 * kXTile + so1) = v1_.l;                               \
  } while (0)

  const f32x16 kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  f32x16 o0, o1;                      // O^T d-blocks 0 / 1: col = query (lane & 31), row = d
#pragma unroll
  for (int r = 0; r < 16; ++r) { o0[r] = 0.f; o1[r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  const float c = 0.125f * 1.44269504088896340736f;      // 1 / sqrt(64) * log2(e)
  constexpr float kDefer = 8.f;

  // loop-invariant LDS byte offsets (see mhsa.hip): K fragment rows l31 (+32: +4096 B), chunk 2 s + hh; V^T fragments through tr_b16
  int koff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) koff[s] = kv_off(l31, 2 * s + hh);
  const int tq = (lane & 15) >> 2, tp = lane & 3, g1 = (lane >> 4) & 1;
  int voff[2][2];
#pragma unroll
  for (int dblk = 0; dblk < 2; ++dblk) {
    const int dcol = dblk * 32 + 16 * g1 + 4 * tp;
    voff[dblk][0] = 2 * kXTile + kv_off(4 * hh + tq, dcol >> 3) + (dcol & 7) * 2;
    voff[dblk][1] = 2 * kXTile + kv_off(4 * hh + tq + 8, dcol >> 3) + (dcol & 7) * 2;
  }
#define SE_XTR(ptr) __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) bf16x4*)(ptr))

  SE_X_ISSUE(0);
  SE_X_WRITE(0);
  __syncthreads();

  for (int kt = 0; kt < nkt; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nkt) SE_X_ISSUE(kt + 1);
    const char* t_s = smem + cur * kXBuf;
    f32x16 s0, s1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 ka_h = *reinterpret_cast<const bf16x8*>(t_s + koff[s]);
      const bf16x8 kb_h = *reinterpret_cast<const bf16x8*>(t_s + koff[s] + 4096);
      const bf16x8 ka_l = *reinterpret_cast<const bf16x8*>(t_s + kXTile + koff[s]);
      const bf16x8 kb_l = *reinterpret_cast<const bf16x8*>(t_s + kXTile + koff[s] + 4096);
      // the two small terms first, the leading term last
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka_l, qh[s], s == 0 ? kZero16 : s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_l, qh[s], s == 0 ? kZero16 : s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka_h, ql[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_h, ql[s], s1, 0, 0, 0);
      s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ka_h, qh[s], s0, 0, 0, 0);
      s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kb_h, qh[s], s1, 0, 0, 0);
    }
    if ((kt + 1) * kAK > len) {
      const int kbase = kt * kAK + 4 * hh;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int key = kbase + (r & 3) + 8 * (r >> 2);
        if (key >= len) s0[r] = -INFINITY;
        if (key + 32 >= len) s1[r] = -INFINITY;
      }
    }
    // exact online softmax (mhsa.hip's exact tile): row maximum over the lane's 32 scores and the 32 of lane ^ 32, deferred rescale
    float mx = fmaxf(s0[0], s1[0]);
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(fmaxf(mx, s0[r]), s1[r]);
    {
      const auto sw_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
      mx = fmaxf(__uint_as_float(sw_[0]), __uint_as_float(sw_[1]));
    }
    const float m_new = ((mx - m_run) * c > kDefer) ? mx : m_run;      // first tile: m_run = -inf -> mx
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
    const float mc = -m_new * c;
    float rs0 = 0.f, rs1 = 0.f;
    bf16x8 ph[2][2], pl[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a0 = __builtin_amdgcn_exp2f(fmaf(s0[8 * s + j], c, mc));
        const float a1 = __builtin_amdgcn_exp2f(fmaf(s1[8 * s + j], c, mc));
        rs0 += a0;
        rs1 += a1;
        const __bf16 h0 = (__bf16)a0, h1 = (__bf16)a1;
        ph[0][s][j] = h0;
        ph[1][s][j] = h1;
        pl[0][s][j] = (__bf16)(a0 - (float)h0);
        pl[1][s][j] = (__bf16)(a1 - (float)h1);
      }
    l_run = fmaf(l_run, alpha, rs0 + rs1);
    m_run = m_new;
    if (__any(alpha != 1.0f)) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
    }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int dblk = 0; dblk < 2; ++dblk) {
          const char* vh_ = t_s + kb * 4096 + s * 2048;
          const bf16x4 lo_h = SE_XTR(vh_ + voff[dblk][0]), hi_h = SE_XTR(vh_ + voff[dblk][1]);
          const bf16x4 lo_l = SE_XTR(vh_ + kXTile + voff[dblk][0]), hi_l = SE_XTR(vh_ + kXTile + voff[dblk][1]);
          const bf16x8 va_h = {lo_h[0], lo_h[1], lo_h[2], lo_h[3], hi_h[0], hi_h[1], hi_h[2], hi_h[3]};
          const bf16x8 va_l = {lo_l[0], lo_l[1], lo_l[2], lo_l[3], hi_l[0], hi_l[1], hi_l[2], hi_l[3]};
          if (dblk == 0) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_l, ph[kb][s], o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_h, pl[kb][s], o0, 0, 0, 0);
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_h, ph[kb][s], o0, 0, 0, 0);
          } else {
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_l, ph[kb][s], o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_h, pl[kb][s], o1, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va_h, ph[kb][s], o1, 0, 0, 0);
          }
        }
      }
    if (kt + 1 < nkt) SE_X_WRITE(cur ^ 1);
    __syncthreads();
  }
#undef SE_X_ISSUE
#undef SE_X_WRITE
#undef SE_XTR

  // ---- epilogue: O / l in fp32; lane holds query q0 + l31, d = 32 dblk + (r & 3) + 8 (r >> 2) + 4 hh
  const float l_tot = l_run + __shfl_xor(l_run, 32);
  const float inv = 1.0f / l_tot;
  const int q = q0 + l31;
  if (q < T && ctx) {
    float* op = ctx + ((size_t)b * T + q) * H + head * kHD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      *reinterpret_cast<float4*>(op + 8 * g) = make_float4(o0[4 * g] * inv, o0[4 * g + 1] * inv, o0[4 * g + 2] * inv, o0[4 * g + 3] * inv);
      *reinterpret_cast<float4*>(op + 32 + 8 * g) = make_float4(o1[4 * g] * inv, o1[4 * g + 1] * inv, o1[4 * g + 2] * inv, o1[4 * g + 3] * inv);
    }
  }
  if (q < T && ctx3) {
    // round 4: the context leaves as the three-term operand [c1 | c1 | c2] of the output projection (se_split3_bf16's activation layout, row stride
    // 3 Kp) -- the separate split pass read the fp32 tensor back and wrote this
    uint16_t* op = ctx3 + ((size_t)b * T + q) * 3 * (size_t)Kp + head * kHD + 4 * hh;
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int blk = 0; blk < 2; ++blk) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (blk ? o1[4 * g + e] : o0[4 * g + e]) * inv;
        uint16_t hi[4], mid[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const __bf16 h = (__bf16)v[e];
          hi[e] = __builtin_bit_cast(uint16_t, h);
          mid[e] = __builtin_bit_cast(uint16_t, (__bf16)(v[e] - (float)h));
        }
        const uint2 ph = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
        const uint2 pm = make_uint2(mid[0] | ((uint32_t)mid[1] << 16), mid[2] | ((uint32_t)mid[3] << 16));
        uint16_t* o = op + 32 * blk + 8 * g;
        *reinterpret_cast<uint2*>(o) = ph;
        *reinterpret_cast<uint2*>(o + Kp) = ph;
        *reinterpret_cast<uint2*>(o + 2 * (size_t)Kp) = pm;
      }
  }
}

}  // namespace se

extern "C" int se_mhsa_fwd_x3_f32(const float* qkv, const int32_t* lengths, int B, int T, int heads, float* ctx, void* stream) {
  SE_REQUIRE(qkv && ctx, "se_mhsa_fwd_x3_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_fwd_x3_f32: bad shape B=%d T=%d heads=%d", B, T, heads);
  SE_REQUIRE((((uintptr_t)qkv | (uintptr_t)ctx) % 16) == 0, "se_mhsa_fwd_x3_f32: buffers must be 16-B aligned");
  const int H = heads * se::kHD;
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  hipLaunchKernelGGL(se::mhsa_x3_kernel, grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, ctx, (uint16_t*)nullptr, 0);
  SE_LAUNCH_CHECK();
  return SE_OK;
}

extern "C" int se_mhsa_fwd_x3_split_f32(const float* qkv, const int32_t* lengths, int B, int T, int heads, uint16_t* ctx3, int Kp, void* stream) {
  SE_REQUIRE(qkv && ctx3, "se_mhsa_fwd_x3_split_f32: null argument");
  SE_REQUIRE(B > 0 && B <= 65535 && T > 0 && heads > 0 && heads <= 65535, "se_mhsa_fwd_x3_split_f32: bad shape B=%d T=%d heads=%d", B, T, heads);
  const int H = heads * se::kHD;
  SE_REQUIRE(Kp == H && Kp % 8 == 0 && (((uintptr_t)qkv | (uintptr_t)ctx3) % 16) == 0, "se_mhsa_fwd_x3_split_f32: Kp = %d must be == %d (no pad columns: they are not written), a multiple of 8, buffers 16-B aligned", Kp, H);
  dim3 grid((T + se::kAQ - 1) / se::kAQ, heads, B);
  hipLaunchKernelGGL(se::mhsa_x3_kernel, grid, dim3(256), 0, se::as_stream(stream), qkv, lengths, T, H, (float*)nullptr, ctx3, Kp);
  SE_LAUNCH_CHECK();
  return SE_OK;
}
