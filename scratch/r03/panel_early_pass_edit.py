# r03 experiment (not shipped): applies the early-pass variant of panelr_kernel to cholesky.hip; A/B in profiles/r03_panel_early_pass_ab.txt
# (build the variants with build.build(variant="emN", defines=("-DSMN_PANEL_EARLY_MFMA=N",)) first)
import re
p='/root/repo/scale-mixtures-of-neural-network-gaussian-processes_amd/csrc/cholesky.hip'
s=open(p).read()
a=s.index("  auto update = [&](int cn, int first) {")
b=s.index("  // products in place, the factored diagonal block to the side buffer)")
# find start of the comment block preceding store_out (line before)
b=s.rfind("\n", 0, s.rfind("\n", 0, b))+1
old=s[a:b]
new='''  // One pass over one or two 16x16 tiles of block column cn: S[t, cn:cn+16] -= S[t, k0:k1] S[cn:cn+16, k0:k1]^T (two
  // independent accumulators, one B fragment).
  auto tile_pass = [&](int ta, int tb, bool two, int k0, int k1, int cn) {
    const int rt[2] = {ta * M::TM, (two ? tb : ta) * M::TM};
    T cv[2][M::ACC];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) cv[j][i] = S[(rt[j] + M::acc_row(lane, i)) * LD + cn + M::acc_col(lane)];
    typename M::acc_t acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int i = 0; i < M::ACC; ++i) acc[j][i] = T(0);
    const T* pb = &S[(cn + fr) * LD + fk];
    const T* pa0 = &S[(rt[0] + fr) * LD + fk];
    const T* pa1 = &S[(rt[1] + fr) * LD + fk];
    // Two fragment sets, ping-pong: the reads of K-step s+1 are issued, THEN the MFMAs of step s (sched_barrier keeps the
    // order; written as a plain prefetch loop hipcc folds it back into read -> wait -> MFMA, 420 cycles per step for 256 of
    // MFMAs).  Reads past the end are clamped to the last step, so no branch sits between the reads and the MFMAs.
    using cvp = const typename M::vec_t*;
    const int klast = k1 - M::KSTEP;
    auto kloop = [&](auto twoc) {
      constexpr bool TWO = decltype(twoc)::value;
      typename M::vec_t b0, x0, y0, b1, x1, y1;
      auto rd = [&](typename M::vec_t& bq, typename M::vec_t& xq, typename M::vec_t& yq, int kb) {
        bq = *reinterpret_cast<cvp>(pb + kb);
        xq = *reinterpret_cast<cvp>(pa0 + kb);
        if (TWO) yq = *reinterpret_cast<cvp>(pa1 + kb);
      };
      auto mm = [&](const typename M::vec_t& bq, const typename M::vec_t& xq, const typename M::vec_t& yq) {
#pragma unroll
        for (int i = 0; i < M::NK; ++i) {
          M::mma1(acc[0], xq[i], bq[i]);
          if (TWO) M::mma1(acc[1], yq[i], bq[i]);
        }
      };
      rd(b0, x0, y0, k0);
      for (int kb = k0;;) {
        rd(b1, x1, y1, min(kb + M::KSTEP, klast));
        __builtin_amdgcn_sched_barrier(0);
        mm(b0, x0, y0);
        kb += M::KSTEP;
        if (kb >= k1) break;
        rd(b0, x0, y0, min(kb + M::KSTEP, klast));
        __builtin_amdgcn_sched_barrier(0);
        mm(b1, x1, y1);
        kb += M::KSTEP;
        if (kb >= k1) break;
      }
    };
    if (two) kloop(std::true_type{}); else kloop(std::false_type{});
#pragma unroll
    for (int j = 0; j < 2; ++j)
      if (j == 0 || two) {
#pragma unroll
        for (int i = 0; i < M::ACC; ++i) S[(rt[j] + M::acc_row(lane, i)) * LD + cn + M::acc_col(lane)] = cv[j][i] - acc[j][i];
      }
  };
  // Block b+1's columns are brought up to date with every finished column BEHIND leaf b, not beside it: f32 MFMAs and the
  // vector instructions of another wave on the same SIMD do not overlap (each MFMA of a partner wave costs the leaf its
  // full 32 cycles, profiles/r03_leaf_probe.txt).  But a helper wave whose SIMD carries NO live row wave (waves w and
  // w + 4 share a SIMD: with six waves helper NW is alone, and a row wave retires once its 64 rows are factored) can
  // apply the part of that update that does not depend on leaf b -- the columns before cb -- beside it for free.  It
  // takes the last `npre(b)` tiles (as many as fit the leaf's duration); behind barrier A those only need K = cb .. cn.
  auto helper_free = [&](int w, int cb) {
    const int p = w - 4;
    return p < 0 || p >= NW || 64 * (p + 1) <= cb;
  };
  auto nfree = [&](int cb) {
    int f = 0;
    for (int w = NW; w < 2 * NW; ++w) f += helper_free(w, cb) ? 1 : 0;
    return f;
  };
  auto npre = [&](int b) {
    if (b == 0 || kPanelEarlyMfma == 0) return 0;
    const int per = kPanelEarlyMfma / (M::NK * (CB / M::KSTEP) * b);   // tiles one free helper finishes beside a leaf
    return min(RT - (b + 1), nfree(b * CB) * per);
  };
  auto early = [&](int b, int fi, int nf) {   // beside leaf b, by free helper fi of nf
    const int cn = (b + 1) * CB, first = b + 1, np = npre(b), base = RT - np;
    (void)first;
    for (int i = fi; i < np; i += 2 * nf) tile_pass(base + i, base + i + nf, i + nf < np, 0, b * CB, cn);
  };
  auto update = [&](int b) {                  // behind barrier A of block b, all 2 NW waves, at most two tiles of a kind each
    const int cn = (b + 1) * CB, first = b + 1, np = npre(b), nn = RT - first - np;
    const int i0 = wave, i1 = wave + 2 * NW;
    if (i0 < nn) tile_pass(first + i0, first + i1, i1 < nn, 0, cn, cn);
    if (i0 < np) tile_pass(first + nn + i0, first + nn + i1, i1 < np, b * CB, cn, cn);
  };
'''
s=s[:a]+new+s[b:]
# helper block
old_h=s[s.index("    // Block b: chunk b+1 (requested three chunks ago) goes to LDS"):s.index("  } else {\n    // ------------------------------------------------------------ row threads")]
new_h='''    // Block b: chunk b+2 (requested three chunks ago) goes to LDS -- one block ahead of its first reader, the early
    // pass beside leaf b+1 -- and chunk b+5 is requested into the same registers.
    auto hblock = [&](int b, auto slotc) {
      constexpr int SLOT = decltype(slotc)::value;
      if (b >= NB) return;
      TL(b, 0);
      if (b + 2 < NB) {
        chunk_store(cbuf[SLOT], b + 2);
        if (b + 5 < NB) chunk_load(cbuf[SLOT], b + 5);
      }
      if (b > 0) store_out(b - 1, hid, NTV);        // beside leaf b
      TL(b, 1);
      if (b + 1 < NB && helper_free(wave, b * CB)) {
        int fi = 0;
        for (int w = NW; w < wave; ++w) fi += helper_free(w, b * CB) ? 1 : 0;
        early(b, fi, nfree(b * CB));
      }
      TL(b, 2);
      if (b + 1 >= NB) return;
      __syncthreads();                              // A: block b is solved in every row
      TL(b, 3);
      update(b);
      TL(b, 4);
      __syncthreads();                              // B: block b + 1 is up to date
      TL(b, 5);
    };
    chunk_load(cbuf[1], 1);
    chunk_load(cbuf[2], 2);
    chunk_load(cbuf[0], 3);
    chunk_store(cbuf[1], 1);                        // read behind barrier A of block 0
    chunk_load(cbuf[1], 4);
    for (int b0 = 0; b0 < NB; b0 += 3) {   // chunk b + 2 lives in slot (b + 2) % 3
      hblock(b0, std::integral_constant<int, 2>{});
      hblock(b0 + 1, std::integral_constant<int, 0>{});
      hblock(b0 + 2, std::integral_constant<int, 1>{});
    }
'''
s=s.replace(old_h,new_h)
s=s.replace("        update(cb + CB, b + 1);\n        TL(b, 4);","        update(b);\n        TL(b, 4);")
open(p,'w').write(s)
