// cnn_resnet.hip — NNGP kernel of the WideResnet of experiments/nt_kernels.py:48-80 (get_conv_resnet_kernel):
//     Conv;  four groups of `block_size` residual blocks, strides 1, 2, 2, 2;  Flatten;  Dense(last_w, b=0)
//     block:  x -> [act; Conv(stride); act; Conv](x) + Shortcut(x),   Shortcut = Conv(stride) in the first block of
//             a group (channel mismatch), Identity in the others;  every Conv is 3x3, SAME, W_std=w, b_std=b.
// The AvgPool of the reference is commented out, so, as for the plain CNN (cnn.hip), only same-pixel covariances
// enter: the state of an image pair is one H x W map that shrinks with the strided convolutions
// (32 -> 32 -> 16 -> 8 -> 4 for CIFAR).  SAME follows lax.padtype_to_pads: out = ceil(in/s),
// pad_total = max((out-1) s + 3 - in, 0), pad_lo = pad_total / 2; the divisor of the window sum stays 9.
//
// Same design as cnn.hip: one wave carries one pair through the whole network on chip; its maps live in LDS with
// a zero ring (three buffers: current, scratch, the block's saved input); the network is a short op list
// interpreted by the kernel; the per-image factor tables r = 1/sqrt(q~) (ReLU) / 1/sqrt(1+2q~) (erf) of every
// activation come from a per-image pass over the same op list on the diagonal.  VALU-bound, no MFMA.
#include "internal.hpp"
#include "nngp_math.hpp"

namespace {

enum : unsigned char { OP_CONV = 0, OP_ACT = 1, OP_SAVE = 2, OP_SCONV = 3, OP_ADD = 4 };
constexpr int kMaxOps = 160;   // 1 + 4 groups * 6 ops * block_size  ->  block_size <= 6

struct NetProg {
  int nops, act, H, W, C;
  int tab_stride;              // factor-table entries per image (sum of h*w over the activations)
  double w2, b2, lw2;
  unsigned char op[kMaxOps], arg[kMaxOps];
};

__host__ __device__ inline int same_out(int in, int s) { return (in + s - 1) / s; }
__host__ __device__ inline int same_lo(int in, int s) {
  const int out = same_out(in, s);
  const int pad = (out - 1) * s + 3 - in;
  return pad > 0 ? pad / 2 : 0;
}

template <typename T>
__device__ __forceinline__ T rcp_t(T x);
template <>
__device__ __forceinline__ float rcp_t<float>(float x) { return __builtin_amdgcn_rcpf(x); }
template <>
__device__ __forceinline__ double rcp_t<double>(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  return fma(fma(-x, r, 1.0), r, r);
}

// A group of NT threads (a workgroup in the per-image pass, one wave in the pair pass) owns three padded maps.
// `sync` orders one op's LDS writes before the next op's reads.
template <typename T, int NT, typename Sync>
struct MapSet {
  T* base;         // three padded maps back to back (no pointer table: buffer i is base + i * PSZ, pure LDS arithmetic)
  int PSZ;         // elements per map, (H0 + 2) * (W0 + 2)
  int PW;          // row stride of every map (W0 + 2)
  int tid;
  Sync sync;

  __device__ __forceinline__ T* b(int i) const { return base + i * PSZ; }

  __device__ __forceinline__ static int free_of(int cur, int skip) {
    for (int i = 0; i < 3; ++i)
      if (i != cur && i != skip) return i;
    return 0;
  }
  __device__ __forceinline__ void zero_ring(T* m, int h, int w) {
    for (int i = tid; i < 2 * (w + 2); i += NT) {
      const int x = i % (w + 2), y = i < w + 2 ? 0 : h + 1;
      m[y * PW + x] = T(0);
    }
    for (int i = tid; i < 2 * h; i += NT) {
      const int y = 1 + i % h, x = i < h ? 0 : w + 1;
      m[y * PW + x] = T(0);
    }
  }
  // dst <- scale * (3x3 SAME window sum of src with stride s) + shift; returns the new dims through oh / ow
  __device__ __forceinline__ void conv(const T* src, T* dst, int h, int w, int s, T scale, T shift, int& oh, int& ow) {
    oh = same_out(h, s);
    ow = same_out(w, s);
    const int lo_h = same_lo(h, s), lo_w = same_lo(w, s);
    const float inv = 1.0f / (float)ow;
    for (int px = tid; px < oh * ow; px += NT) {
      const int y = (int)(((float)px + 0.5f) * inv), x = px - y * ow;
      const T* c = src + (s * y - lo_h + 1) * PW + (s * x - lo_w + 1);   // top-left of the window, padded coordinates
      const T bs = ((c[0] + c[1]) + (c[2] + c[PW])) + ((c[PW + 1] + c[PW + 2]) + (c[2 * PW] + c[2 * PW + 1])) + c[2 * PW + 2];
      dst[(y + 1) * PW + x + 1] = fma(scale, bs, shift);
    }
    zero_ring(dst, oh, ow);
  }
  __device__ __forceinline__ void add(T* into, const T* from, int h, int w) {
    const float inv = 1.0f / (float)w;
    for (int px = tid; px < h * w; px += NT) {
      const int y = (int)(((float)px + 0.5f) * inv), x = px - y * w;
      into[(y + 1) * PW + x + 1] += from[(y + 1) * PW + x + 1];
    }
  }
};

struct BlockSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};
struct WaveSync {
  __device__ __forceinline__ void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
};

// ---------------------------------------------------------------- per image: variance maps, factor tables, diagonal
template <typename T>
__global__ void __launch_bounds__(256) resnet_q_kernel(const T* __restrict__ x, NetProg p, T* __restrict__ R,
                                                       T* __restrict__ diag) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = p.W + 2, PSZ = (p.H + 2) * PW;
  double* base = reinterpret_cast<double*>(smem);
  MapSet<double, 256, BlockSync> ms{base, PSZ, PW, (int)threadIdx.x, BlockSync{}};
  double* red = base + 3 * PSZ;
  const int64_t img = blockIdx.x;
  for (int i = threadIdx.x; i < 3 * PSZ; i += 256) base[i] = 0.0;
  __syncthreads();
  int h = p.H, w = p.W, cur = 0, skip = -1, toff = 0;
  for (int px = threadIdx.x; px < h * w; px += 256) {
    const T* xp = x + (img * h * w + px) * p.C;
    double s = 0.0;
    for (int c = 0; c < p.C; ++c) s += (double)xp[c] * (double)xp[c];
    ms.b(0)[(px / w + 1) * PW + px % w + 1] = s / p.C;
  }
  __syncthreads();
  for (int o = 0; o < p.nops; ++o) {
    const int op = p.op[o], arg = p.arg[o];
    if (op == OP_CONV) {
      const int dst = ms.free_of(cur, skip);
      int oh, ow;
      ms.conv(ms.b(cur), ms.b(dst), h, w, arg, p.w2 / 9.0, p.b2, oh, ow);
      cur = dst; h = oh; w = ow;
    } else if (op == OP_ACT) {
      const int dst = ms.free_of(cur, skip);
      for (int px = threadIdx.x; px < h * w; px += 256) {
        const int y = px / w, xx = px % w;
        const double q = ms.b(cur)[(y + 1) * PW + xx + 1];
        double r, qa;
        if (p.act == 0) {
          r = q > 0.0 ? 1.0 / sqrt(q) : 0.0;
          qa = 0.5 * q;
        } else {
          r = 1.0 / sqrt(1.0 + 2.0 * q);
          qa = (2.0 / nngp::kPi) * asin(2.0 * q / (1.0 + 2.0 * q));
        }
        R[img * p.tab_stride + toff + px] = (T)r;
        ms.b(dst)[(y + 1) * PW + xx + 1] = qa;
      }
      ms.zero_ring(ms.b(dst), h, w);
      toff += h * w;
      cur = dst;
    } else if (op == OP_SAVE) {
      skip = cur;
    } else if (op == OP_SCONV) {
      const int dst = ms.free_of(cur, skip);
      int oh, ow;
      ms.conv(ms.b(skip), ms.b(dst), h * arg, w * arg, arg, p.w2 / 9.0, p.b2, oh, ow);   // the saved input has the pre-stride dims
      __syncthreads();
      ms.add(ms.b(cur), ms.b(dst), h, w);
      skip = -1;
    } else {   // OP_ADD
      ms.add(ms.b(cur), ms.b(skip), h, w);
      skip = -1;
    }
    __syncthreads();
  }
  double s = 0.0;
  for (int px = threadIdx.x; px < h * w; px += 256) s += ms.b(cur)[(px / w + 1) * PW + px % w + 1];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) diag[img] = (T)(p.lw2 * red[0] / (h * w));
}

// ---------------------------------------------------------------- per pair
template <typename T>
struct RPairArgs {
  const T* x1; const T* x2; const T* R1; const T* R2; const T* diag;
  int64_t n1, n2; int symmetric, mirror;
  NetProg prog;
  T* out; int64_t ldo; int64_t npairs;
};

template <typename T, int ACT>
__global__ void __launch_bounds__(256) resnet_pair_kernel(RPairArgs<T> a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const NetProg& p = a.prog;
  const int PW = p.W + 2, PSZ = (p.H + 2) * PW;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T* base = reinterpret_cast<T*>(smem) + (size_t)wave * 3 * PSZ;
  MapSet<T, 64, WaveSync> ms{base, PSZ, PW, lane, WaveSync{}};
  for (int i = lane; i < 3 * PSZ; i += 64) base[i] = T(0);
  const T scale = (T)(p.w2 / 9.0), shift = (T)p.b2, inv_c = (T)(1.0 / p.C);
  const int64_t stride = (int64_t)gridDim.x * 4;
  for (int64_t pr = (int64_t)blockIdx.x * 4 + wave; pr < a.npairs; pr += stride) {
    int64_t n, m;
    if (a.symmetric) {
      int64_t r = (int64_t)((sqrt(8.0 * (double)pr + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= pr) ++r;
      while (r * (r + 1) / 2 > pr) --r;
      n = r;
      m = pr - r * (r + 1) / 2;
    } else {
      n = pr / a.n2;
      m = pr % a.n2;
    }
    ms.sync();
    int h = p.H, w = p.W, cur = 0, skip = -1, toff = 0;
    const T* xa = a.x1 + n * h * w * p.C;
    const T* xb = a.x2 + m * h * w * p.C;
    // channel loop outside, NB pixels inside: 2 NB loads in flight per channel (one dependent L2 round trip per pixel
    // and channel made this phase pure latency: cnn.hip, profiles/r01f_notes.md).  fp32: +14 % on the whole kernel;
    // fp64 lives at its register limit and loses 3 % with a batch of 8, so it keeps one pixel at a time.
    constexpr int NB = sizeof(T) == 8 ? 1 : 8;
    for (int px0 = lane; px0 < h * w; px0 += 64 * NB) {
      T s[NB];
#pragma unroll
      for (int j = 0; j < NB; ++j) s[j] = T(0);
      for (int c = 0; c < p.C; ++c) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int pc = min(px0 + 64 * j, h * w - 1);
          s[j] = fma(xa[pc * p.C + c], xb[pc * p.C + c], s[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int px = px0 + 64 * j;
        if (px < h * w) ms.b(0)[(px / w + 1) * PW + px % w + 1] = s[j] * inv_c;
      }
    }
    ms.zero_ring(ms.b(0), h, w);
    ms.sync();
    const T* r1 = a.R1 + n * p.tab_stride;
    const T* r2 = a.R2 + m * p.tab_stride;
    for (int o = 0; o < p.nops; ++o) {
      const int op = p.op[o], arg = p.arg[o];
      if (op == OP_CONV) {
        const int dst = ms.free_of(cur, skip);
        int oh, ow;
        ms.conv(ms.b(cur), ms.b(dst), h, w, arg, scale, shift, oh, ow);
        cur = dst; h = oh; w = ow;
      } else if (op == OP_ACT) {
        const int dst = ms.free_of(cur, skip);
        const float inv = 1.0f / (float)w;
        for (int px = lane; px < h * w; px += 64) {
          const int y = (int)(((float)px + 0.5f) * inv), xx = px - y * w;
          const T k = ms.b(cur)[(y + 1) * PW + xx + 1];
          const T rr = r1[toff + px] * r2[toff + px];
          T kn;
          if (ACT == 0) {
            const T ss = rr > T(0) ? T(1.0 / (2.0 * nngp::kPi)) * rcp_t<T>(rr) : T(0);
            kn = nngp::relu_map<T, false>(k, rr, ss).k;
          } else {
            kn = nngp::erf_map<T, false>(k, rr, T(0)).k;
          }
          ms.b(dst)[(y + 1) * PW + xx + 1] = kn;
        }
        ms.zero_ring(ms.b(dst), h, w);
        toff += h * w;
        cur = dst;
      } else if (op == OP_SAVE) {
        skip = cur;
      } else if (op == OP_SCONV) {
        const int dst = ms.free_of(cur, skip);
        int oh, ow;
        ms.conv(ms.b(skip), ms.b(dst), h * arg, w * arg, arg, scale, shift, oh, ow);
        ms.sync();
        ms.add(ms.b(cur), ms.b(dst), h, w);
        skip = -1;
      } else {
        ms.add(ms.b(cur), ms.b(skip), h, w);
        skip = -1;
      }
      ms.sync();
    }
    T s = T(0);
    for (int px = lane; px < h * w; px += 64) s += ms.b(cur)[(px / w + 1) * PW + px % w + 1];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
      T v = (T)p.lw2 * s / (T)(h * w);
      if (a.symmetric && n == m) v = a.diag[n];
      a.out[n * a.ldo + m] = v;
      if (a.symmetric && a.mirror && n != m) a.out[m * a.ldo + n] = v;
    }
  }
}

// WideResnet(block_size, k = 1) as an op list; returns the factor-table length per image (or -1: too many ops,
// -2: a strided stage whose input size is not a multiple of the stride -- the shortcut bookkeeping assumes it is).
int build_prog(NetProg* p, int block_size, int H, int W) {
  int n = 0, h = H, w = W, tab = 0;
  auto push = [&](unsigned char op, unsigned char arg) {
    if (n < kMaxOps) { p->op[n] = op; p->arg[n] = arg; }
    ++n;
  };
  push(OP_CONV, 1);
  const int strides[4] = {1, 2, 2, 2};
  for (int g = 0; g < 4; ++g)
    for (int b = 0; b < block_size; ++b) {
      const int s = b == 0 ? strides[g] : 1;
      if (s > 1 && (h % s || w % s)) return -2;
      push(OP_SAVE, 0);
      push(OP_ACT, 0); tab += h * w;
      push(OP_CONV, (unsigned char)s);
      h = same_out(h, s); w = same_out(w, s);
      push(OP_ACT, 0); tab += h * w;
      push(OP_CONV, 1);
      if (b == 0) push(OP_SCONV, (unsigned char)s);
      else push(OP_ADD, 0);
    }
  if (n > kMaxOps) return -1;
  p->nops = n;
  p->tab_stride = tab;
  return tab;
}

template <typename T>
int resnet_t(smn_ctx* ctx, int act, int block_size, double w, double b, double lw, const void* x1, int64_t n1,
             const void* x2, int64_t n2, int64_t H, int64_t W, int64_t C, int fill, void* out, int64_t ldk) {
  const bool sym = x2 == nullptr;
  if (sym) n2 = n1;
  NetProg p{};
  p.act = act; p.H = (int)H; p.W = (int)W; p.C = (int)C;
  p.w2 = w * w; p.b2 = b * b; p.lw2 = lw * lw;
  const int tab = build_prog(&p, block_size, (int)H, (int)W);
  if (tab == -1) return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_conv_resnet: block_size %d needs more than %d ops", block_size, kMaxOps);
  if (tab == -2) return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_conv_resnet: image size %lldx%lld is not divisible by the strides (needs multiples of 8)", (long long)H, (long long)W);
  const size_t psz = (size_t)(H + 2) * (W + 2);
  const size_t lds_q = (3 * psz + 256) * sizeof(double);
  const size_t lds_p = 4 * 3 * psz * sizeof(T);
  if (lds_q > 160 * 1024 || lds_p > 160 * 1024)
    return smn_fail(ctx, SMN_ENOTSUP, "smn_kernel_conv_resnet: image %lldx%lld too large for the on-chip pair maps", (long long)H, (long long)W);
  const size_t qn1 = (size_t)n1 * (size_t)tab, qn2 = sym ? 0 : (size_t)n2 * (size_t)tab;
  void* tv = nullptr;
  SMN_TRY(smn_workspace(ctx, 1, sizeof(T) * (qn1 + qn2 + (size_t)n1 + (size_t)n2), &tv));
  T* R1 = static_cast<T*>(tv);
  T* R2 = sym ? R1 : R1 + qn1;
  T* d1 = R1 + qn1 + qn2;
  T* d2 = d1 + n1;
  {
    ProfScope ps(ctx, PROF_PREP, ctx->stream);
    SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(resnet_q_kernel<T>), lds_q));
    hipLaunchKernelGGL(resnet_q_kernel<T>, dim3((unsigned)n1), dim3(256), lds_q, ctx->stream, static_cast<const T*>(x1), p, R1, d1);
    if (!sym)
      hipLaunchKernelGGL(resnet_q_kernel<T>, dim3((unsigned)n2), dim3(256), lds_q, ctx->stream, static_cast<const T*>(x2), p, R2, d2);
  }
  SMN_CHECK_LAUNCH(ctx);
  RPairArgs<T> a;
  a.x1 = static_cast<const T*>(x1); a.x2 = sym ? a.x1 : static_cast<const T*>(x2);
  a.R1 = R1; a.R2 = R2; a.diag = d1; a.n1 = n1; a.n2 = n2;
  a.symmetric = sym ? 1 : 0; a.mirror = (sym && fill == SMN_FILL_FULL) ? 1 : 0;
  a.prog = p; a.out = static_cast<T*>(out); a.ldo = ldk;
  a.npairs = sym ? n1 * (n1 + 1) / 2 : n1 * n2;
  int64_t blocks = (a.npairs + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  {
    ProfScope ps(ctx, PROF_BUILD, ctx->stream);
    if (act == SMN_ACT_RELU) {
      SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(resnet_pair_kernel<T, 0>), lds_p));
      hipLaunchKernelGGL((resnet_pair_kernel<T, 0>), dim3((unsigned)blocks), dim3(256), lds_p, ctx->stream, a);
    } else {
      SMN_TRY(smn_allow_lds(ctx, reinterpret_cast<const void*>(resnet_pair_kernel<T, 1>), lds_p));
      hipLaunchKernelGGL((resnet_pair_kernel<T, 1>), dim3((unsigned)blocks), dim3(256), lds_p, ctx->stream, a);
    }
  }
  SMN_CHECK_LAUNCH(ctx);
  return SMN_OK;
}

}  // namespace

extern "C" int smn_kernel_conv_resnet(smn_ctx* ctx, int dtype, int act, int block_size, double w_std, double b_std,
                                      double last_w_std, const void* x1_d, int64_t n1, const void* x2_d, int64_t n2,
                                      int64_t H, int64_t W, int64_t C, int fill, void* nngp_d, int64_t ldk) {
  if (!ctx || !x1_d || !nngp_d) return SMN_EINVAL;
  SMN_ENTER(ctx);
  if (dtype != SMN_F32 && dtype != SMN_F64) return smn_fail(ctx, SMN_EINVAL, "bad dtype %d", dtype);
  if (act != SMN_ACT_RELU && act != SMN_ACT_ERF) return smn_fail(ctx, SMN_EINVAL, "Unsupported act %d", act);
  if (n1 <= 0 || (x2_d && n2 <= 0) || H <= 0 || W <= 0 || C <= 0 || block_size <= 0)
    return smn_fail(ctx, SMN_EINVAL, "smn_kernel_conv_resnet: bad sizes");
  if (dtype == SMN_F64)
    return resnet_t<double>(ctx, act, block_size, w_std, b_std, last_w_std, x1_d, n1, x2_d, n2, H, W, C, fill, nngp_d, ldk);
  return resnet_t<float>(ctx, act, block_size, w_std, b_std, last_w_std, x1_d, n1, x2_d, n2, H, W, C, fill, nngp_d, ldk);
}
