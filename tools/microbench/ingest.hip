// Per-CU operand ingest rate on gfx950: how fast can ONE 512-thread workgroup per CU pull bytes from L2 / HBM
//   mode 0: LDS-DMA (buffer_load ... lds, 16 B per lane), counted vmcnt, K steps in flight
//   mode 1: buffer_load_dwordx4 into registers, consumed by a cheap v_xor (no LDS)
//   mode 2: buffer_load_dwordx4 into registers + ds_write_b128 (register staging)
//   mode 3: mode 0 with an EMPTY descriptor (no memory traffic at all: instruction cost only)
// Each workgroup streams `bytes_per_wg` from its own region (set `shared` to make all workgroups of an XCD read the same
// region = L2 hits).  hipcc --offload-arch=gfx950 -O3 -o ingest ingest.hip && ./ingest
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int MODE, int PER_THREAD, int INFLIGHT>
__global__ __launch_bounds__(512) void ingest_kernel(const unsigned char *src, long long bytes_per_wg, int shared, unsigned *sink) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const long long region = shared ? (long long)(blockIdx.x & 7) * bytes_per_wg : (long long)blockIdx.x * bytes_per_wg;
  const unsigned char *base = src + region;
  constexpr int kStepB = 512 * 16 * PER_THREAD;            // bytes per step and workgroup
  const int n_steps = (int)(bytes_per_wg / kStepB);
  u32x4 acc = {0u, 0u, 0u, 0u};
  if (MODE == 0 || MODE == 3) {
    auto issue = [&](int st) {
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base + (long long)(st < n_steps ? st : 0) * kStepB), 0,
                                                                         (MODE == 3 || st >= n_steps) ? 0 : kStepB, 0x00020000);
      unsigned char *slot = smem + (st % 4) * 32768;
#pragma unroll
      for (int j = 0; j < PER_THREAD; ++j)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + (tid - lane + 512 * j) * 16), 16,
                                                 (unsigned)(tid + 512 * j) * 16u, 0, 0, 0);
    };
    for (int st = 0; st < INFLIGHT; ++st) issue(st);
    for (int st = 0; st < n_steps; ++st) {
      issue(st + INFLIGHT);
      if (INFLIGHT * PER_THREAD == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (INFLIGHT * PER_THREAD == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (INFLIGHT * PER_THREAD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (INFLIGHT * PER_THREAD == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    acc.x = *reinterpret_cast<unsigned *>(smem + tid * 4);
  } else {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, (int)bytes_per_wg, 0x00020000);
    u32x4 r[INFLIGHT][PER_THREAD];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k)
#pragma unroll
      for (int j = 0; j < PER_THREAD; ++j) r[k][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(k * kStepB + (tid + 512 * j) * 16), 0, 0);
    for (int st = 0; st < n_steps; st += INFLIGHT) {
#pragma unroll
      for (int k = 0; k < INFLIGHT; ++k) {
        u32x4 cur[PER_THREAD];
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) cur[j] = r[k][j];
        const unsigned off = (unsigned)((st + k + INFLIGHT) * kStepB);
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) r[k][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + (unsigned)(tid + 512 * j) * 16u, 0, 0);
        if (MODE == 2) {
          unsigned char *slot = smem + ((st + k) % 4) * 32768;
#pragma unroll
          for (int j = 0; j < PER_THREAD; ++j) *reinterpret_cast<u32x4 *>(slot + (tid + 512 * j) * 16) = cur[j];
          __builtin_amdgcn_s_barrier();
        } else {
#pragma unroll
          for (int j = 0; j < PER_THREAD; ++j) acc ^= cur[j];
        }
      }
    }
    if (MODE == 2) { __syncthreads(); acc.x ^= *reinterpret_cast<unsigned *>(smem + tid * 4); }
  }
  if (acc.x == 0x12345678u && acc.y == 77u) sink[0] = acc.z + acc.w;
}

// mode P: the A-image pattern of csrc/wgrad_wide_bf16.cuh -- each step is 32 (x PT) rows of 256 B taken out of rows of
// `ld_bytes` (a column tile of a row-major matrix); piece i = tid + 512 j: row i / 16, 16-byte chunk (i % 16) ^ swz(row) when
// SWZ, else i % 16.  All workgroups of an XCD read the same rows when `shared`.
__global__ __launch_bounds__(512) void pattern_kernel(const unsigned char *src, int ld_bytes, int n_steps, int shared, long long region_bytes,
                                                      unsigned *sink, int per_thread, int swz_on) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const unsigned char *base = src + (shared ? (long long)(blockIdx.x & 7) : (long long)blockIdx.x) * region_bytes;
  unsigned voff[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i = tid + 512 * j, r = i >> 4, sl = i & 15;
    const int swz = ((r & 3) << 2) | ((r >> 2) & 3);
    voff[j] = j < per_thread ? (unsigned)r * (unsigned)ld_bytes + (unsigned)((swz_on ? sl ^ swz : sl) * 16) : 0x80000000u;
  }
  const long long step_adv = (long long)32 * per_thread * ld_bytes;
  const int step_bytes = (32 * per_thread - 1) * ld_bytes + 256;
  auto issue = [&](int st) {          // always 4 instructions (the surplus ones out of range)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base + (long long)(st < n_steps ? st : 0) * step_adv), 0,
                                                                       st >= n_steps ? 0 : step_bytes, 0x00020000);
    unsigned char *slot = smem + (st % 4) * 32768;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void *)(slot + (tid - lane + 512 * j) * 16), 16, voff[j], 0, 0, 0);
  };
  issue(0); issue(1);
  for (int st = 0; st < n_steps; ++st) {
    issue(st + 2);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (*reinterpret_cast<unsigned *>(smem + tid * 4) == 0x12345678u) sink[0] = 1;
}
void run_pattern(int PT, int SWZ, const unsigned char *src, int ld_bytes, int shared, unsigned *sink, int wgs, long long region_bytes) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int n_steps = (int)(region_bytes / ((long long)32 * PT * ld_bytes));
  for (int i = 0; i < 3; ++i) pattern_kernel<<<dim3(wgs), dim3(512), 0, 0>>>(src, ld_bytes, n_steps, shared, region_bytes, sink, PT, SWZ);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) pattern_kernel<<<dim3(wgs), dim3(512), 0, 0>>>(src, ld_bytes, n_steps, shared, region_bytes, sink, PT, SWZ);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, bytes = (double)n_steps * 32 * PT * 256, gbs_cu = bytes / (us * 1e-6) / 1e9;
  printf("{\"mode\": \"lds_dma_tile_rows256B\", \"pieces_per_thread\": %d, \"xor_swizzle\": %d, \"row_stride_B\": %d, \"shared_l2\": %d, \"steps\": %d, \"us\": %.1f, \"GBps_per_CU\": %.1f, \"TBps_chip\": %.2f}\n",
         PT, SWZ, ld_bytes, shared, n_steps, us, gbs_cu, gbs_cu * wgs / 1e3);
}

template <int MODE, int PT, int INF> void run(const char *name, const unsigned char *src, long long bytes_per_wg, int shared, unsigned *sink, int wgs) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((ingest_kernel<MODE, PT, INF>), dim3(wgs), dim3(512), 0, 0, src, bytes_per_wg, shared, sink);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ingest_kernel<MODE, PT, INF>), dim3(wgs), dim3(512), 0, 0, src, bytes_per_wg, shared, sink);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, gbs_cu = bytes_per_wg / (us * 1e-6) / 1e9;
  printf("{\"mode\": \"%s\", \"per_thread\": %d, \"inflight_steps\": %d, \"shared_l2\": %d, \"wgs\": %d, \"MB_per_wg\": %.2f, \"us\": %.1f, \"GBps_per_CU\": %.1f, \"TBps_chip\": %.2f}\n",
         name, PT, INF, shared, wgs, bytes_per_wg / 1e6, us, gbs_cu, gbs_cu * wgs / 1e3);
}

int main() {
  const int wgs = 256;
  const long long per_wg = 2LL << 20;          // 2 MiB per workgroup: 512 MiB in all (beyond the 256 MiB infinity cache)
  unsigned char *src; unsigned *sink;
  CHECK(hipMalloc(&src, per_wg * wgs)); CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemset(src, 1, per_wg * wgs));
  for (int shared = 0; shared < 2; ++shared) {
    run<0, 4, 2>("lds_dma", src, per_wg, shared, sink, wgs);
    run<0, 4, 3>("lds_dma", src, per_wg, shared, sink, wgs);
    run<0, 2, 2>("lds_dma", src, per_wg, shared, sink, wgs);
    run<0, 2, 4>("lds_dma", src, per_wg, shared, sink, wgs);
    run<3, 4, 2>("lds_dma_empty_descriptor", src, per_wg, shared, sink, wgs);
    run<1, 4, 2>("regs", src, per_wg, shared, sink, wgs);
    run<1, 4, 4>("regs", src, per_wg, shared, sink, wgs);
    run<1, 2, 4>("regs", src, per_wg, shared, sink, wgs);
    run<2, 4, 2>("regs_ds_write", src, per_wg, shared, sink, wgs);
    run<2, 4, 4>("regs_ds_write", src, per_wg, shared, sink, wgs);
  }
  const int lds[3] = {256, 768, 2048};
  for (int shared = 0; shared < 2; ++shared)
    for (int k = 0; k < 3; ++k) {
      const int ld = lds[k];
      run_pattern(1, 0, src, ld, shared, sink, wgs, per_wg);
      run_pattern(1, 1, src, ld, shared, sink, wgs, per_wg);
      run_pattern(4, 0, src, ld, shared, sink, wgs, per_wg);
      run_pattern(4, 1, src, ld, shared, sink, wgs, per_wg);
    }
  return 0;
}
