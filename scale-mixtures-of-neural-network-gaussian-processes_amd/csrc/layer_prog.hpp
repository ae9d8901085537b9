// layer_prog.hpp — the layer stack of experiments/nt_kernels.py as a per-element program.
//
//   MLP (nt_kernels.py:21-31):      L x [Dense(w,b); act];  Dense(last_w, b=0)
//   dense ResNet (nt_kernels.py:83-103):  Dense(w,b);  L x {K <- [act; Dense(w,b)](K) + K};  act;  Dense(last_w,0)
//
// Dense:  K <- w^2 K + b^2,  Theta <- K_new + w^2 Theta      (NTK parameterisation)
// act:    (K, Kdot) <- map(K; r_i r_j, s_i s_j),  Theta <- Theta * Kdot      (A.2)
//
// One "set" = one activation with its per-row tables; MLP has L sets, ResNet L + 1.
//
// FAST mode (f32, MLP, ReLU, NNGP only — the configuration BASELINE.json is quoted on): the recursion
// runs in CORRELATION space, which folds each Dense into the table products:
//     x_0 = K0;   c_l = clamp(x_l * (u^l_i u^l_j) + v^l_i v^l_j);   x_{l+1} = J(c_l);   K = x_L * (sigma_i sigma_j)
//     u^0 = w r^0, v^l = b r^l, u^l = w s^{l-1} r^l (l >= 1), sigma = last_w s^{L-1},  r = 1/sqrt(q~), s = sqrt(q~/2pi)
// (the code carries x / pi: J comes back divided by pi and s is sqrt(q~/2) -- one multiplication less per layer)
// It is the same arithmetic (c_l is exactly the argument the generic map forms) with 5 ops less per layer
// and a single-sqrt J (nngp_math.hpp).  The table slots hold (u, v) instead of (r, s) and sigma rides in
// the NTK-diagonal slot; diag_tables_kernel writes whichever the flag asks for.
// Erf (f32, MLP, NNGP only; round 4) takes the same form: x_{l+1} = asin(c_l) with the same table recipe on r' = sqrt(2) r
// and the constant s = sqrt(2/pi) (u^0 = w r', u^l = w s r', v = b r', sigma = last_w s), and a one-branch asin
// (nngp_math.hpp asin_fast): 21.2 -> 13 vector instructions per element and layer (profiles/r04_recursion_instruction_budget.txt).
// Round 3 had tried the folding alone (4 instructions of ~24) and measured no gain; with the shorter asin the map-bound
// depths do move (profiles/r04_recursion_table.txt).  asin is steep at |c| -> 1, so entries of near-duplicate rows differ
// from the generic path by up to 1e-4 in f32 -- both are equally far from the fp64 value there.
#pragma once
#include "nngp_math.hpp"

enum { NET_MLP = 0, NET_RESNET = 1, NET_NONE = 2 };
enum { ACT_RELU = 0, ACT_ERF = 1 };
constexpr int kMaxSets = 16;

struct LayerProg {  // plain-old-data kernel argument
  int net, act, nsets;
  int fast;         // tables are (u, v, sigma): must equal ElemProg<...>::FAST of the kernel launched
  double w2, b2, lw2;
};

template <typename T, int NET, int ACT, bool NTK>
struct ElemProg {
  static constexpr bool FAST = sizeof(T) == 4 && NET == NET_MLP && !NTK;   // ReLU and erf
  T w2, b2, lw2;
  int nsets;
  __device__ __forceinline__ explicit ElemProg(const LayerProg& p)
      : w2((T)p.w2), b2((T)p.b2), lw2((T)p.lw2), nsets(p.nsets) {}

  static __device__ __forceinline__ nngp::ActOut<T> act(T k, T rr, T ss) {
    if (ACT == ACT_RELU) return nngp::relu_map<T, NTK>(k, rr, ss);
    return nngp::erf_map<T, NTK>(k, rr, ss);
  }
  __device__ __forceinline__ void pre(T& k, T& th) const {
    if (NET == NET_RESNET) {
      k = fma(w2, k, b2);
      if (NTK) th = k;
    } else if (NTK) {
      th = T(0);
    }
  }
  // rr / ss: products of the two per-row table entries of this set (r_i r_j, s_i s_j; FAST: u_i u_j, v_i v_j)
  __device__ __forceinline__ void step(int set, T& k, T& th, T rr, T ss) const {
    if constexpr (FAST) {
      const float c = fmaf(k, rr, ss);   // (clamped where it matters: inside the maps)
      k = ACT == ACT_RELU ? nngp::relu_j_fast(c) : nngp::asin_fast(c);
    } else if (NET == NET_MLP) {
      const T kt = fma(w2, k, b2);
      T tht = T(0);
      if (NTK) tht = fma(w2, th, kt);
      const nngp::ActOut<T> o = act(kt, rr, ss);
      k = o.k;
      if (NTK) th = tht * o.kdot;
    } else if (NET == NET_RESNET) {
      const nngp::ActOut<T> o = act(k, rr, ss);
      if (set == nsets - 1) {
        k = o.k;
        if (NTK) th *= o.kdot;
      } else {
        const T ka = fma(w2, o.k, b2);
        if (NTK) th += fma(w2, th * o.kdot, ka);
        k += ka;
      }
    }
  }
  // sig = sigma_i sigma_j (FAST only; ignored otherwise)
  __device__ __forceinline__ void post(T& k, T& th, T sig) const {
    if constexpr (FAST) {
      k *= sig;
    } else if (NET != NET_NONE) {
      k *= lw2;
      if (NTK) th = fma(lw2, th, k);
    }
  }
};
