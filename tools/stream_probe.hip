// Read-stream probe (round 3): what does the ACCESS SHAPE of a once-read weight stream cost on MI355X?
//   frag   the MFMA-fragment shape of gemm_skinny_keepx_kernel: a wave-instruction reads 16 rows x 64 B (half lines), the
//          other half of each 128-B line comes with the next instruction of the same wave
//   line   every wave-instruction reads 1 KiB contiguous (8 whole 128-B lines)
//   dma    the same 1 KiB pieces by LDS-DMA (global_load_lds_dwordx4) into a ring in LDS, nothing consumes them
// each with non-temporal or default-policy loads, 512-thread workgroups, one per CU (256) or two, over a buffer far larger
// than the Infinity Cache (rotating 262 MB windows of a 2 GiB buffer, back to back) and once from cold caches.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/stream_probe tools/stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ u32x4 ld16(const u32x4* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}

// rows of 8 KiB (K = 4096 bf16); a block owns `rows_per_block` consecutive rows.  Wave w, super-step it: bytes [256*(w+8*it), +256)
// of 16 rows; instruction s: 64 B of each row.  Exactly the keepx kernel's addresses (RT = 1, no rotation).
template <bool NT, int DEPTH>
__global__ __launch_bounds__(512) void frag_kernel(const char* __restrict__ W, int rows_per_block, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, g = lane >> 4;
  const char* base = W + (size_t)blockIdx.x * rows_per_block * 8192;
  u32x4 acc = {0, 0, 0, 0};
  const int n_ts = rows_per_block / 16;
  // software pipeline: DEPTH super-steps (4 loads each) in flight
  u32x4 buf[DEPTH][4];
  auto addr = [&](int step, int s) {
    const int ts = step >> 2, it = step & 3;
    return reinterpret_cast<const u32x4*>(base + (size_t)(ts * 16 + r16) * 8192 + (wave + 8 * it) * 256 + s * 64 + g * 16);
  };
  const int total = n_ts * 4;
#pragma unroll
  for (int d = 0; d < DEPTH; ++d)
#pragma unroll
    for (int s = 0; s < 4; ++s) buf[d][s] = ld16<NT>(addr(d < total ? d : total - 1, s));
  for (int step = 0; step < total; step += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
      for (int s = 0; s < 4; ++s) acc ^= buf[d][s];
      const int nx = step + d + DEPTH;
#pragma unroll
      for (int s = 0; s < 4; ++s) buf[d][s] = ld16<NT>(addr(nx < total ? nx : total - 1, s));
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// every wave-instruction 1 KiB contiguous; a block owns bytes_per_block contiguous bytes, its 8 waves interleave 1 KiB pieces
template <bool NT, int DEPTH>
__global__ __launch_bounds__(512) void line_kernel(const char* __restrict__ W, size_t bytes_per_block, unsigned* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = W + (size_t)blockIdx.x * bytes_per_block + lane * 16;
  const int total = (int)(bytes_per_block / (8 * 1024));     // pieces per wave
  u32x4 acc = {0, 0, 0, 0};
  u32x4 buf[DEPTH];
  auto addr = [&](int i) { return reinterpret_cast<const u32x4*>(base + ((size_t)i * 8 + wave) * 1024); };
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) buf[d] = ld16<NT>(addr(d < total ? d : total - 1));
  for (int i = 0; i < total; i += DEPTH) {
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      acc ^= buf[d];
      const int nx = i + d + DEPTH;
      buf[d] = ld16<NT>(addr(nx < total ? nx : total - 1));
    }
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[0] = 1;
}

// LDS-DMA: each wave owns a ring of SLOTS x 1 KiB in LDS and keeps SLOTS pieces in flight; nothing reads them
template <int AUX, int SLOTS>
__global__ __launch_bounds__(512) void dma_kernel(const char* __restrict__ W, size_t bytes_per_block, unsigned* out) {
  extern __shared__ __attribute__((aligned(1024))) char ring[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const char* base = W + (size_t)blockIdx.x * bytes_per_block + lane * 16;
  const int total = (int)(bytes_per_block / (8 * 1024));
  char* my = ring + wave * SLOTS * 1024;
  auto issue = [&](int i, int slot) {
    const char* src = base + ((size_t)(i < total ? i : total - 1) * 8 + wave) * 1024;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src),
                                     (__attribute__((address_space(3))) void*)(my + slot * 1024), 16, 0, AUX);
  };
#pragma unroll
  for (int d = 0; d < SLOTS; ++d) issue(d, d);
  for (int i = 0; i < total; i += SLOTS) {
#pragma unroll
    for (int d = 0; d < SLOTS; ++d) {
      // wait until the oldest piece has landed (SLOTS - 1 may stay in flight), then reuse its slot
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SLOTS - 1) : "memory");
      issue(i + d + SLOTS, d);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (ring[threadIdx.x] == 0x7f && out) out[1] = 1;
}

int main(int argc, char** argv) {
  const size_t MAT = 32000ull * 8192;                  // 262 MB: the Llama-2 lm_head
  const size_t BIG = 8 * MAT;                          // 2.1 GB: rotating windows defeat the 256 MB Infinity Cache
  char* buf; unsigned* out;
  CHECK(hipMalloc(&buf, BIG)); CHECK(hipMalloc(&out, 64));
  CHECK(hipMemset(buf, 0x11, BIG)); CHECK(hipMemset(out, 0, 64));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipStream_t st = 0;
  struct V { const char* name; int id; };
  for (int wgs_per_cu = 1; wgs_per_cu <= 2; ++wgs_per_cu) {
    const int grid = 250 * wgs_per_cu;                 // 32000 rows / 250 = 128 rows (1 MiB) per block at 1 per CU
    const int rpb = 32000 / grid;
    const size_t bpb = (size_t)rpb * 8192;
    auto run = [&](int id, const char* w) {
      switch (id) {
        case 0: hipLaunchKernelGGL((frag_kernel<true, 2>), dim3(grid), dim3(512), 0, st, w, rpb, out); break;
        case 1: hipLaunchKernelGGL((frag_kernel<true, 4>), dim3(grid), dim3(512), 0, st, w, rpb, out); break;
        case 2: hipLaunchKernelGGL((frag_kernel<false, 2>), dim3(grid), dim3(512), 0, st, w, rpb, out); break;
        case 3: hipLaunchKernelGGL((line_kernel<true, 8>), dim3(grid), dim3(512), 0, st, w, bpb, out); break;
        case 4: hipLaunchKernelGGL((line_kernel<true, 16>), dim3(grid), dim3(512), 0, st, w, bpb, out); break;
        case 5: hipLaunchKernelGGL((line_kernel<false, 8>), dim3(grid), dim3(512), 0, st, w, bpb, out); break;
        case 6: hipLaunchKernelGGL((dma_kernel<2, 8>), dim3(grid), dim3(512), 8 * 8 * 1024, st, w, bpb, out); break;
        case 7: hipLaunchKernelGGL((dma_kernel<0, 8>), dim3(grid), dim3(512), 8 * 8 * 1024, st, w, bpb, out); break;
        case 8: hipLaunchKernelGGL((dma_kernel<2, 16>), dim3(grid), dim3(512), 8 * 16 * 1024, st, w, bpb, out); break;
      }
    };
    const char* names[] = {"frag nt depth2 (keepx shape)", "frag nt depth4", "frag default depth2", "line nt depth8", "line nt depth16",
                           "line default depth8", "dma nt 8 slots", "dma default 8 slots", "dma nt 16 slots"};
    for (int round = 0; round < 2; ++round)
      for (int id = 0; id < 9; ++id) {
        // back to back over rotating windows (steady state), 16 launches
        for (int i = 0; i < 4; ++i) run(id, buf + (size_t)(i % 8) * MAT);
        CHECK(hipEventRecord(e0, st));
        for (int i = 0; i < 16; ++i) run(id, buf + (size_t)(i % 8) * MAT);
        CHECK(hipEventRecord(e1, st));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // single launches behind a different kernel's window (dependent-launch shape): time each alone
        std::vector<float> singles;
        for (int i = 0; i < 6; ++i) {
          run((id + 3) % 6, buf + (size_t)((i + 4) % 8) * MAT);      // something else streams another window first
          CHECK(hipEventRecord(e0, st));
          run(id, buf + (size_t)(i % 8) * MAT);
          CHECK(hipEventRecord(e1, st));
          CHECK(hipEventSynchronize(e1));
          float m1; CHECK(hipEventElapsedTime(&m1, e0, e1));
          singles.push_back(m1);
        }
        std::sort(singles.begin(), singles.end());
        if (round == 1)
          printf("%d WG/CU  %-30s back-to-back %6.1f us/launch = %5.2f TB/s | single launch median %6.1f us = %5.2f TB/s (min %6.1f)\n", wgs_per_cu,
                 names[id], ms / 16 * 1e3, MAT / (ms / 16 * 1e-3) / 1e12, singles[3] * 1e3, MAT / (singles[3] * 1e-3) / 1e12, singles[0] * 1e3);
      }
  }
  CHECK(hipDeviceSynchronize());
  unsigned h[2]; CHECK(hipMemcpy(h, out, 8, hipMemcpyDeviceToHost));
  printf("done %u %u\n", h[0], h[1]);
  return 0;
}
