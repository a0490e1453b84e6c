// Row exchange of the class-table gradient for data-parallel training (reference: train.py:152,257 -- DDP all-reduces the
// (num_classes + 1) x D table gradient densely, 162 MB for DiT-B, although only the step's <= B label rows are non-zero).
// Each rank contributes its labels SORTED plus one copy of every touched row (duplicates zeroed) -- osud_table_rows_pack --,
// the host all-gathers both arrays, and every rank rebuilds the touched rows by summing the contributions in RANK ORDER --
// osud_table_rows_apply -- so replicas stay bit-identical (and equal to what a rank-ordered dense sum gives).
// Two launches per side instead of the ~25 small tensor-library launches (sort, masks, index_fill / index_add per rank) the
// host code needed; sits on the exposed tail of the exchange, after the last backward phase.
#include "kernels.h"

namespace osud {
namespace {

constexpr int kMaxLabels = 4096;  // labels per sort (B per rank; W * B for the union)

// bitonic sort of n <= NP (power of two) int64 keys held in LDS, ascending; 256 threads
template <int NP> __device__ void bitonic_sort(long long* keys) {
  for (int k = 2; k <= NP; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int i = threadIdx.x; i < NP; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k) == 0;
          const long long a = keys[i], b = keys[l];
          if ((a > b) == up) {
            keys[i] = b;
            keys[l] = a;
          }
        }
      }
    }
  __syncthreads();
}

// every block sorts the (few hundred) labels itself -- cheaper than a second launch -- then block i emits entry i:
//   idx_out[i] = i-th smallest label, rows_out[i] = the table-gradient row of that label if it is the label's first occurrence, else 0
template <int NP>
__global__ __launch_bounds__(256) void table_rows_pack_kernel(const float* __restrict__ table_grad, int rows, int D,
                                                              const int64_t* __restrict__ labels, int B, int64_t* __restrict__ idx_out,
                                                              float* __restrict__ rows_out) {
  __shared__ long long keys[NP];
  for (int i = threadIdx.x; i < NP; i += blockDim.x) {
    long long v = 0x7fffffffffffffffLL;
    if (i < B) {
      v = labels[i];
      v = v < 0 ? 0 : (v >= rows ? rows - 1 : v);  // as the forward clamps (cond_kernel)
    }
    keys[i] = v;
  }
  bitonic_sort<NP>(keys);
  const int i = blockIdx.x;
  const long long lbl = keys[i];
  const bool first = i == 0 || keys[i - 1] != lbl;
  if (threadIdx.x == 0) idx_out[i] = lbl;
  const float* src = table_grad + (size_t)lbl * D;
  float* dst = rows_out + (size_t)i * D;
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    const float4 v = first ? *reinterpret_cast<const float4*>(src + d) : make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(dst + d) = v;
  }
}

// one block: the sorted union of all ranks' labels without duplicates -> uniq[0 .. n), n -> count[0]
template <int NP>
__global__ __launch_bounds__(256) void table_rows_union_kernel(const int64_t* __restrict__ all_idx, int total, int64_t* __restrict__ uniq,
                                                               int* __restrict__ count) {
  __shared__ long long keys[NP];
  __shared__ int n_out;
  for (int i = threadIdx.x; i < NP; i += blockDim.x) keys[i] = i < total ? (long long)all_idx[i] : 0x7fffffffffffffffLL;
  if (threadIdx.x == 0) n_out = 0;
  bitonic_sort<NP>(keys);
  // compaction in order: one thread walks the (<= 4096) sorted keys -- order matters only for reproducible addresses, not values
  if (threadIdx.x == 0) {
    int n = 0;
    for (int i = 0; i < total; ++i)
      if (i == 0 || keys[i] != keys[i - 1]) uniq[n++] = keys[i];
    n_out = n;
    count[0] = n;
  }
}

// block j < count: label L = uniq[j]; table_grad[L] = sum over ranks r = 0 .. W-1 (in that order) of rank r's row for L
// (binary search in its sorted list; a rank's duplicates carry zero rows, the first occurrence the data)
__global__ __launch_bounds__(256) void table_rows_apply_kernel(float* __restrict__ table_grad, int D, const int64_t* __restrict__ all_idx,
                                                               const float* __restrict__ all_rows, int W, int B,
                                                               const int64_t* __restrict__ uniq, const int* __restrict__ count) {
  if ((int)blockIdx.x >= count[0]) return;
  const long long L = uniq[blockIdx.x];
  __shared__ int pos[64];  // first occurrence of L in rank r's list, or -1
  if ((int)threadIdx.x < W) {
    const int64_t* lst = all_idx + (size_t)threadIdx.x * B;
    int lo = 0, hi = B;  // lower bound
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (lst[mid] < L) lo = mid + 1;
      else hi = mid;
    }
    pos[threadIdx.x] = (lo < B && lst[lo] == L) ? lo : -1;
  }
  __syncthreads();
  for (int d = threadIdx.x * 4; d < D; d += blockDim.x * 4) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int r = 0; r < W; ++r) {
      if (pos[r] < 0) continue;
      const float4 v = *reinterpret_cast<const float4*>(all_rows + ((size_t)r * B + pos[r]) * D + d);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    *reinterpret_cast<float4*>(table_grad + (size_t)L * D + d) = acc;
  }
}

}  // namespace
}  // namespace osud

using namespace osud;

extern "C" int osud_table_rows_pack(const float* table_grad, int rows, int D, const int64_t* labels, int B, int64_t* idx_out,
                                    float* rows_out, osud_stream stream) {
  OSUD_CHECK_ARG(table_grad && labels && idx_out && rows_out && rows > 0 && B > 0 && D > 0 && D % 4 == 0,
                 "table_rows_pack: bad argument");
  OSUD_CHECK_ARG(B <= kMaxLabels, "table_rows_pack: at most %d labels per rank (got %d)", kMaxLabels, B);
  hipStream_t st = (hipStream_t)stream;
  if (B <= 256) hipLaunchKernelGGL((table_rows_pack_kernel<256>), dim3(B), dim3(256), 0, st, table_grad, rows, D, labels, B, idx_out, rows_out);
  else if (B <= 1024) hipLaunchKernelGGL((table_rows_pack_kernel<1024>), dim3(B), dim3(256), 0, st, table_grad, rows, D, labels, B, idx_out, rows_out);
  else hipLaunchKernelGGL((table_rows_pack_kernel<kMaxLabels>), dim3(B), dim3(256), 0, st, table_grad, rows, D, labels, B, idx_out, rows_out);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

extern "C" int osud_table_rows_apply(float* table_grad, int rows, int D, const int64_t* all_idx, const float* all_rows, int W, int B,
                                     int64_t* scratch, osud_stream stream) {
  OSUD_CHECK_ARG(table_grad && all_idx && all_rows && scratch && rows > 0 && W > 0 && W <= 64 && B > 0 && D % 4 == 0,
                 "table_rows_apply: bad argument");
  const int total = W * B;
  OSUD_CHECK_ARG(total <= kMaxLabels, "table_rows_apply: at most %d labels over all ranks (got %d x %d)", kMaxLabels, W, B);
  hipStream_t st = (hipStream_t)stream;
  int64_t* uniq = scratch;                                // [total]
  int* count = reinterpret_cast<int*>(scratch + total);   // [1] (+ padding): scratch holds W * B + 1 int64
  if (total <= 256) hipLaunchKernelGGL((table_rows_union_kernel<256>), dim3(1), dim3(256), 0, st, all_idx, total, uniq, count);
  else if (total <= 1024) hipLaunchKernelGGL((table_rows_union_kernel<1024>), dim3(1), dim3(256), 0, st, all_idx, total, uniq, count);
  else hipLaunchKernelGGL((table_rows_union_kernel<kMaxLabels>), dim3(1), dim3(256), 0, st, all_idx, total, uniq, count);
  OSUD_HIP(hipGetLastError());
  hipLaunchKernelGGL(table_rows_apply_kernel, dim3(total), dim3(256), 0, st, table_grad, D, all_idx, all_rows, W, B, uniq, count);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}
