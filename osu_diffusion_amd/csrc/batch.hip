// Batched housekeeping kernels: one launch for a whole LIST of small buffers.
// A training step re-packs ~70 weights (fp32 master -> TE and TE^T), copies ~40 small fp32 vectors and zeroes
// ~55 gradient accumulators; as individual hipMemsetAsync / hipMemcpyAsync / convert / transpose launches these
// were ~250 dispatches of 3-6 us each (7 % of the step).  A list travels by value in the kernel arguments.
#include "kernels.h"

namespace osud {

namespace {

template <int OP, typename TE> __global__ __launch_bounds__(256) void seg_kernel(SegList L) {
  const int s = blockIdx.y;
  const size_t n = L.n[s];
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  if (OP == SEG_ZERO) {  // n = 16-byte chunks
    uint4* d = reinterpret_cast<uint4*>(L.dst[s]);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) d[i] = make_uint4(0, 0, 0, 0);
  } else if (OP == SEG_COPY || sizeof(TE) == 4) {  // n = 16-byte chunks (4 floats)
    const uint4* a = reinterpret_cast<const uint4*>(L.src[s]);
    uint4* d = reinterpret_cast<uint4*>(L.dst[s]);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) d[i] = a[i];
  } else {  // SEG_CONVERT to bf16: n = groups of 4 elements (16 bytes in, 8 bytes out)
    const float4* a = reinterpret_cast<const float4*>(L.src[s]);
    uint2* d = reinterpret_cast<uint2*>(L.dst[s]);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += stride) {
      const float4 v = a[i];
      d[i] = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    }
  }
}

// out[c][r] = in[r][c] for every matrix of the list: 64x64 tiles through LDS; block b finds its matrix by its tile range
template <typename TE> __global__ __launch_bounds__(256) void transpose_many_kernel(TransposeList L) {
  __shared__ float tile[64][65];
  int s = 0;
  while (s + 1 < L.count && (int)blockIdx.x >= L.tile_begin[s + 1]) ++s;
  const int t = blockIdx.x - L.tile_begin[s];
  const int R = L.R[s], C = L.C[s], tc = C / 64;
  const int r0 = (t / tc) * 64, c0 = (t % tc) * 64;
  const TE* in = reinterpret_cast<const TE*>(L.src[s]);
  TE* out = reinterpret_cast<TE*>(L.dst[s]);
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int r = ty + 4 * i;
    tile[r][tx] = load_elem(in + (size_t)(r0 + r) * C + c0 + tx);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = ty + 4 * i;
    store_elem(out + (size_t)(c0 + c) * R + r0 + tx, tile[tx][c]);
  }
}

}  // namespace

int launch_segments(int op, int prec, const SegList& L, hipStream_t st) {
  if (L.count == 0) return OSUD_OK;
  OSUD_CHECK_ARG(L.count <= SegList::kMax, "segments: list too long (%d)", L.count);
  size_t nmax = 0;
  for (int i = 0; i < L.count; ++i) nmax = L.n[i] > nmax ? L.n[i] : nmax;
  int gx = (int)((nmax + 1023) / 1024);  // ~4 items per thread
  gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
  const dim3 grid(gx, L.count);
  if (op == SEG_ZERO) hipLaunchKernelGGL((seg_kernel<SEG_ZERO, float>), grid, dim3(256), 0, st, L);
  else if (op == SEG_COPY) hipLaunchKernelGGL((seg_kernel<SEG_COPY, float>), grid, dim3(256), 0, st, L);
  else if (prec == OSUD_PREC_BF16) hipLaunchKernelGGL((seg_kernel<SEG_CONVERT, bf16_t>), grid, dim3(256), 0, st, L);
  else hipLaunchKernelGGL((seg_kernel<SEG_CONVERT, float>), grid, dim3(256), 0, st, L);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

int launch_transpose_many(int prec, TransposeList& L, hipStream_t st) {
  if (L.count == 0) return OSUD_OK;
  OSUD_CHECK_ARG(L.count <= TransposeList::kMax, "transpose_many: list too long (%d)", L.count);
  int tiles = 0;
  for (int i = 0; i < L.count; ++i) {
    OSUD_CHECK_ARG(L.R[i] % 64 == 0 && L.C[i] % 64 == 0, "transpose_many: %d x %d is not a multiple of 64", L.R[i], L.C[i]);
    L.tile_begin[i] = tiles;
    tiles += (L.R[i] / 64) * (L.C[i] / 64);
  }
  if (prec == OSUD_PREC_BF16) hipLaunchKernelGGL((transpose_many_kernel<bf16_t>), dim3(tiles), dim3(256), 0, st, L);
  else hipLaunchKernelGGL((transpose_many_kernel<float>), dim3(tiles), dim3(256), 0, st, L);
  OSUD_HIP(hipGetLastError());
  return OSUD_OK;
}

}  // namespace osud
