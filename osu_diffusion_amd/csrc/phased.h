// What the phased main loops (gemm_phased.h, wgrad.hip) share: the arithmetic of a phase table and its compile-time checks.
//
// A table S describes how one K stage of a tile is cut into NPH phases for a workgroup whose two waves per SIMD run one barrier
// apart (the design is written up at the top of gemm_phased.h):
//   PPW            1 KiB LDS-DMA pieces per wave and stage ("slots" 0 .. PPW - 1, issued in slot order: the staging stream)
//   AHEAD          pieces per wave in flight in front of the consumer when a stage's phase 0 begins
//   cnt[p]         pieces per wave issued in phase p                                   (sum = PPW)
//   need[p]        highest slot the fragment reads of phase p touch, -1 if it reads nothing
//   read_phase[m]  the phase that reads slot m's LDS region (the last one, if several do)
// gfx950 only.
#pragma once
#include <utility>

namespace osud {
namespace {

template <typename S> constexpr int ph_issued_before(int p) {  // pieces per wave issued since the slab's phase 0, before phase p
  int n = 0;
  for (int q = 0; q < p; ++q) n += S::cnt[q];
  return n;
}
// the counted wait of phase p: how many of this wave's pieces may still be in flight when phase p + 1's reads must have landed
template <typename S> constexpr int ph_wait(int p) {
  const int nxt = (p + 1) % S::NPH;
  if (S::need[nxt] < 0) return -1;
  const int needpos = (p + 1 == S::NPH ? S::PPW : 0) + S::need[nxt];
  return S::AHEAD + ph_issued_before<S>(p + 1) - (needpos + 1);
}
// ... and whether that wait, in the FIRST stage of a tile, only covers pieces issued before the tile began -- all of which the wait
// in front of the previous tile's epilogue (or the prologue) has retired already, so it can be skipped (it would otherwise sit behind
// the epilogue's stores, which share the counter)
template <typename S> constexpr bool ph_wait_predrained(int p) {
  const int nxt = (p + 1) % S::NPH;
  if (S::need[nxt] < 0) return true;
  return (p + 1 == S::NPH ? S::PPW : 0) + S::need[nxt] < S::AHEAD;
}
template <typename S> constexpr bool sched_ok() {
  int total = 0;
  for (int p = 0; p < S::NPH; ++p) total += S::cnt[p];
  if (total != S::PPW) return false;  // the stream advances one slab per slab
  for (int p = 0; p < S::NPH; ++p) {
    if (S::need[(p + 1) % S::NPH] >= 0 && ph_wait<S>(p) < 0) return false;  // RAW: the needed piece must have been ISSUED by then
    for (int i = 0; i < S::cnt[p]; ++i) {
      const int pos = S::AHEAD + ph_issued_before<S>(p) + i, d = pos / S::PPW, m = pos % S::PPW;
      if (d < 1 || d > 2) return false;                        // two buffers: the slab after this one, or the one after that
      if (d == 2 && p < S::read_phase[m] + 2) return false;    // WAR: same buffer as the slab being consumed
      if (d == 1 && p + S::NPH < S::read_phase[m] + 2) return false;
    }
  }
  return true;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"i"(N) : "memory"); }

template <class F, int... I> __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

}  // namespace
}  // namespace osud
