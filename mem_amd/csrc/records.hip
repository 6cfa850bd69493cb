// Event record contract: raw dataset records -> the (N,4) float64 rows
// [x, y, t, p] that every later stage of the path reads.  HBM-bound byte shuffling: each record is read once
// (5 B / a few columns) and one 32-byte row is written with two 16-byte stores per lane.
//   * N-Caltech101 5-byte records      process_data/process_dataset.py:48-63
//   * N-ImageNet column arrays          mem/dataset_folder.py:285-292 (imgnet_npy_loader)
//   * DSEC rows (p <- 2p-1, y < 440)    mem/dataset_folder.py:275-283 (dsec_npy_loader)
#include "common.h"

namespace {

using namespace memhip;

constexpr int kT = 256;

// byte0 -> column 0, byte1 -> column 1, polarity = bit 7 of byte2 -> 2p-1, t = (byte2 & 0x7f):byte3:byte4 big-endian.
// A workgroup owns 256 records = 1280 bytes, fetched as 320 aligned dwords into LDS (coalesced), then one record
// per lane.
__global__ __launch_bounds__(kT) void decode_ncaltech_kernel(const uint8_t* __restrict__ raw, long long n,
                                                             double* __restrict__ ev) {
  __shared__ uint32_t sm[kT * 5 / 4];
  const long long r0 = (long long)blockIdx.x * kT;
  const long long nrec = n - r0 < kT ? n - r0 : kT;
  const long long nbytes = nrec * 5;
  const uint8_t* src = raw + r0 * 5;                      // r0*5 = blockIdx*1280: dword aligned when raw is
  for (int w = threadIdx.x; w < kT * 5 / 4; w += kT) {
    const long long b = (long long)w * 4;
    uint32_t v = 0;
    if (b + 4 <= nbytes) {
      v = *reinterpret_cast<const uint32_t*>(src + b);
    } else {
      for (int k = 0; k < 4; ++k)
        if (b + k < nbytes) v |= (uint32_t)src[b + k] << (8 * k);
    }
    sm[w] = v;
  }
  __syncthreads();
  if (threadIdx.x < nrec) {
    const uint8_t* q = reinterpret_cast<const uint8_t*>(sm) + threadIdx.x * 5;
    const uint32_t b0 = q[0], b1 = q[1], b2 = q[2], b3 = q[3], b4 = q[4];
    const uint32_t t = ((b2 & 0x7fu) << 16) | (b3 << 8) | b4;
    double2* o = reinterpret_cast<double2*>(ev + (r0 + threadIdx.x) * 4);
    o[0] = make_double2((double)b0, (double)b1);
    o[1] = make_double2((double)t, 2.0 * (double)((b2 >> 7) & 1u) - 1.0);
  }
}

// dtype codes of memhip_events_from_columns / memhip_events_dsec (include/memhip.h)
__device__ __forceinline__ double load_as_f64(const void* p, int dt, long long i) {
  switch (dt) {
    case MEMHIP_DT_U8: case MEMHIP_DT_BOOL: return (double)reinterpret_cast<const uint8_t*>(p)[i];
    case MEMHIP_DT_I8: return (double)reinterpret_cast<const int8_t*>(p)[i];
    case MEMHIP_DT_U16: return (double)reinterpret_cast<const uint16_t*>(p)[i];
    case MEMHIP_DT_I16: return (double)reinterpret_cast<const int16_t*>(p)[i];
    case MEMHIP_DT_U32: return (double)reinterpret_cast<const uint32_t*>(p)[i];
    case MEMHIP_DT_I32: return (double)reinterpret_cast<const int32_t*>(p)[i];
    case MEMHIP_DT_U64: return (double)reinterpret_cast<const unsigned long long*>(p)[i];
    case MEMHIP_DT_I64: return (double)reinterpret_cast<const long long*>(p)[i];
    case MEMHIP_DT_F32: return (double)reinterpret_cast<const float*>(p)[i];
    default: return reinterpret_cast<const double*>(p)[i];
  }
}

// low byte of the polarity column as int8 (ndarray.astype(np.int8) of an integer / bool array truncates)
__device__ __forceinline__ int8_t load_low_i8(const void* p, int dt, long long i) {
  switch (dt) {
    case MEMHIP_DT_U8: case MEMHIP_DT_BOOL: case MEMHIP_DT_I8: return reinterpret_cast<const int8_t*>(p)[i];
    case MEMHIP_DT_U16: case MEMHIP_DT_I16: return (int8_t)reinterpret_cast<const uint16_t*>(p)[i];
    case MEMHIP_DT_U32: case MEMHIP_DT_I32: return (int8_t)reinterpret_cast<const uint32_t*>(p)[i];
    default: return (int8_t)reinterpret_cast<const unsigned long long*>(p)[i];
  }
}

// ps = p.astype(int8) * 2 - 1 in int8 arithmetic (wraps), rows = [x, y, t, ps] as float64
__global__ __launch_bounds__(kT) void events_from_columns_kernel(const void* __restrict__ x, int dx, const void* __restrict__ y,
                                                                 int dy, const void* __restrict__ t, int dtt,
                                                                 const void* __restrict__ p, int dp, long long n,
                                                                 double* __restrict__ ev) {
  for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
    int8_t q = load_low_i8(p, dp, i);
    q = (int8_t)(q * 2);
    q = (int8_t)(q - 1);
    double2* o = reinterpret_cast<double2*>(ev + i * 4);
    o[0] = make_double2(load_as_f64(x, dx, i), load_as_f64(y, dy, i));
    o[1] = make_double2(load_as_f64(t, dtt, i), (double)q);
  }
}

// ---- DSEC: rows with y < y_limit survive, in order (stream compaction: count -> scan -> scatter)
constexpr int kRowsPerBlock = 1024;

__device__ __forceinline__ bool dsec_keep(const void* in, int dt, long long r, double y_limit) {
  return load_as_f64(in, dt, r * 4 + 1) < y_limit;
}

__global__ __launch_bounds__(kT) void dsec_count_kernel(const void* __restrict__ in, int dt, long long n, double y_limit,
                                                        long long* __restrict__ block_count) {
  __shared__ int sc[kT / 64];
  const long long r0 = (long long)blockIdx.x * kRowsPerBlock;
  int c = 0;
  for (int k = threadIdx.x; k < kRowsPerBlock; k += kT) {
    const long long r = r0 + k;
    if (r < n && dsec_keep(in, dt, r, y_limit)) ++c;
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
  if ((threadIdx.x & 63) == 0) sc[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_count[blockIdx.x] = sc[0] + sc[1] + sc[2] + sc[3];
}

// exclusive scan of the block counts in place (one workgroup; nb is a few thousand at most), total -> n_out
__global__ __launch_bounds__(kT) void dsec_scan_kernel(long long* __restrict__ block_count, int nb, long long* __restrict__ n_out) {
  __shared__ long long part[kT];
  const int per = (nb + kT - 1) / kT;
  const int b0 = threadIdx.x * per;
  long long s = 0;
  for (int k = 0; k < per; ++k)
    if (b0 + k < nb) s += block_count[b0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long run = 0;
    for (int i = 0; i < kT; ++i) { const long long v = part[i]; part[i] = run; run += v; }
    *n_out = run;
  }
  __syncthreads();
  long long run = part[threadIdx.x];
  for (int k = 0; k < per; ++k)
    if (b0 + k < nb) { const long long v = block_count[b0 + k]; block_count[b0 + k] = run; run += v; }
}

__global__ __launch_bounds__(kT) void dsec_scatter_kernel(const void* __restrict__ in, int dt, long long n, double y_limit,
                                                          const long long* __restrict__ block_offset,
                                                          double* __restrict__ out) {
  __shared__ int wave_base[kT / 64];
  __shared__ int chunk_base;
  const long long r0 = (long long)blockIdx.x * kRowsPerBlock;
  if (threadIdx.x == 0) chunk_base = 0;
  __syncthreads();
  const long long dst0 = block_offset[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k0 = 0; k0 < kRowsPerBlock; k0 += kT) {         // rows in order: chunk by chunk, wave by wave, lane by lane
    const long long r = r0 + k0 + threadIdx.x;
    const bool keep = r < n && dsec_keep(in, dt, r, y_limit);
    const unsigned long long bal = __ballot(keep);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_base[wave] = __popcll(bal);
    __syncthreads();
    int base = chunk_base;
    for (int w = 0; w < wave; ++w) base += wave_base[w];
    if (keep) {
      double2* o = reinterpret_cast<double2*>(out + (dst0 + base + before) * 4);
      o[0] = make_double2(load_as_f64(in, dt, r * 4), load_as_f64(in, dt, r * 4 + 1));
      o[1] = make_double2(load_as_f64(in, dt, r * 4 + 2), 2.0 * load_as_f64(in, dt, r * 4 + 3) - 1.0);
    }
    __syncthreads();
    if (threadIdx.x == 0) chunk_base += wave_base[0] + wave_base[1] + wave_base[2] + wave_base[3];
    __syncthreads();
  }
}

bool dtype_ok(int dt) { return dt >= MEMHIP_DT_U8 && dt <= MEMHIP_DT_BOOL; }
bool int_dtype(int dt) { return dt != MEMHIP_DT_F32 && dt != MEMHIP_DT_F64 && dtype_ok(dt); }

}  // namespace

extern "C" int memhip_decode_ncaltech101(const uint8_t* raw, int64_t n_bytes, double* ev, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n_bytes >= 0, "decode_ncaltech101: negative size");
  if (n_bytes % 5 != 0) return fail(MEMHIP_EINVAL, "decode_ncaltech101: truncated record (%lld bytes is not a multiple of 5)",
                                    (long long)n_bytes);
  if (n_bytes == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(raw && ev && ((uintptr_t)raw & 3) == 0 && ((uintptr_t)ev & 15) == 0,
                 "decode_ncaltech101: raw must be 4-byte and ev 16-byte aligned");
  const long long n = n_bytes / 5;
  hipLaunchKernelGGL(decode_ncaltech_kernel, dim3((unsigned)((n + kT - 1) / kT)), dim3(kT), 0, as_stream(stream), raw, n, ev);
  return check_launch("decode_ncaltech101");
}

extern "C" int memhip_events_from_columns(const void* x, int x_dtype, const void* y, int y_dtype, const void* t, int t_dtype,
                                          const void* p, int p_dtype, int64_t n, double* ev, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0, "events_from_columns: negative size");
  if (n == 0) return MEMHIP_OK;
  MEMHIP_REQUIRE(x && y && t && p && ev && ((uintptr_t)ev & 15) == 0, "events_from_columns: null / unaligned argument");
  MEMHIP_REQUIRE(dtype_ok(x_dtype) && dtype_ok(y_dtype) && dtype_ok(t_dtype), "events_from_columns: unknown dtype code");
  if (!int_dtype(p_dtype)) return fail(MEMHIP_EUNSUPPORTED, "events_from_columns: polarity column must be bool / integer");
  long long nb = (n + kT - 1) / kT;
  if (nb > 8192) nb = 8192;
  hipLaunchKernelGGL(events_from_columns_kernel, dim3((unsigned)nb), dim3(kT), 0, as_stream(stream), x, x_dtype, y, y_dtype, t,
                     t_dtype, p, p_dtype, (long long)n, ev);
  return check_launch("events_from_columns");
}

extern "C" size_t memhip_events_dsec_workspace(int64_t n) {
  return (size_t)((n + kRowsPerBlock - 1) / kRowsPerBlock + 1) * sizeof(long long);
}

extern "C" int memhip_events_dsec(const void* in, int dtype, int64_t n, double y_limit, double* out, int64_t* n_out,
                                  void* workspace, size_t workspace_bytes, memhip_stream_t stream) {
  MEMHIP_REQUIRE(n >= 0 && n_out, "events_dsec: bad arguments");
  hipStream_t s = as_stream(stream);
  if (n == 0) {
    MEMHIP_HIP(hipMemsetAsync(n_out, 0, sizeof(int64_t), s));
    return MEMHIP_OK;
  }
  MEMHIP_REQUIRE(in && out && dtype_ok(dtype) && ((uintptr_t)out & 15) == 0, "events_dsec: null / unaligned argument");
  if (workspace_bytes < memhip_events_dsec_workspace(n)) return fail(MEMHIP_EWORKSPACE, "events_dsec: workspace too small");
  const int nb = (int)((n + kRowsPerBlock - 1) / kRowsPerBlock);
  long long* cnt = reinterpret_cast<long long*>(workspace);
  hipLaunchKernelGGL(dsec_count_kernel, dim3(nb), dim3(kT), 0, s, in, dtype, (long long)n, y_limit, cnt);
  hipLaunchKernelGGL(dsec_scan_kernel, dim3(1), dim3(kT), 0, s, cnt, nb, reinterpret_cast<long long*>(n_out));
  hipLaunchKernelGGL(dsec_scatter_kernel, dim3(nb), dim3(kT), 0, s, in, dtype, (long long)n, y_limit, cnt, out);
  return check_launch("events_dsec");
}
