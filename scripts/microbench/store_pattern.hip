// Store-tail microbenchmark: one 512-thread workgroup per CU writes a 256x256 bf16 tile (128 KiB) of a row-major
// [M][N] matrix with different lane->address patterns (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int PAT>
__global__ __launch_bounds__(512) void k(char* C, long ldc_bytes, int tiles_n, unsigned long long* stamps) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, wr = w >> 2, wc = w & 3;
  const int tm = blockIdx.x / tiles_n, tn = blockIdx.x % tiles_n;
  char* base = C + (long)tm * 256 * ldc_bytes + (long)tn * 512;
  uint4 v = make_uint4(tid, lane, w, 1);
  unsigned long long t0 = wall_clock64();
  if (PAT == 0) {            // frag_out: per instruction 16 rows x 64 B; wave tile 128 rows x 128 B
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int jp = 0; jp < 2; ++jp)
        *reinterpret_cast<uint4*>(base + (long)(wr * 128 + i * 16 + lr) * ldc_bytes + wc * 128 + jp * 64 + (lg & 1) * 32 + (lg >> 1) * 16) = v;
  } else if (PAT == 1) {     // per instruction 8 rows x 128 B (full lines); wave tile 128 rows x 128 B
#pragma unroll
    for (int it = 0; it < 16; ++it)
      *reinterpret_cast<uint4*>(base + (long)(wr * 128 + it * 8 + (lane >> 3)) * ldc_bytes + wc * 128 + (lane & 7) * 16) = v;
  } else if (PAT == 2) {     // per instruction 2 rows x 512 B: wave w owns rows w*32 .. +31 of the whole 256-column tile
#pragma unroll
    for (int it = 0; it < 16; ++it)
      *reinterpret_cast<uint4*>(base + (long)(w * 32 + it * 2 + (lane >> 5)) * ldc_bytes + (lane & 31) * 16) = v;
  } else if (PAT == 3) {     // as 0 with dwordx2 halves (32 stores)
    const int lr = lane & 15, lg = lane >> 4;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint2*>(base + (long)(wr * 128 + i * 16 + lr) * ldc_bytes + wc * 128 + j * 32 + lg * 8) = make_uint2(v.x, v.y);
  }
  unsigned long long t1 = wall_clock64();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  unsigned long long t2 = wall_clock64();
  if (tid == 0) { stamps[blockIdx.x * 3] = t0; stamps[blockIdx.x * 3 + 1] = t1; stamps[blockIdx.x * 3 + 2] = t2; }
}

int main() {
  const int shapes[3][2] = {{25088, 512}, {4096, 4096}, {50176, 512}};
  for (auto& sh : shapes) {
    const long M = sh[0], N = sh[1];
    char* C; unsigned long long* st;
    CK(hipMalloc(&C, M * N * 2)); CK(hipMalloc(&st, 4096 * 3 * 8));
    const int tiles_n = N / 256, nt = (M / 256) * tiles_n;
    for (int pat = 0; pat < 4; ++pat) {
      std::vector<unsigned long long> h(nt * 3);
      double iss = 0, tot = 0;
      for (int rep = 0; rep < 3; ++rep) {
        switch (pat) {
          case 0: hipLaunchKernelGGL(k<0>, dim3(nt), dim3(512), 0, 0, C, N * 2, tiles_n, st); break;
          case 1: hipLaunchKernelGGL(k<1>, dim3(nt), dim3(512), 0, 0, C, N * 2, tiles_n, st); break;
          case 2: hipLaunchKernelGGL(k<2>, dim3(nt), dim3(512), 0, 0, C, N * 2, tiles_n, st); break;
          default: hipLaunchKernelGGL(k<3>, dim3(nt), dim3(512), 0, 0, C, N * 2, tiles_n, st); break;
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), st, nt * 3 * 8, hipMemcpyDeviceToHost));
        iss = tot = 0;
        unsigned long long tmin = ~0ull, tmax = 0;
        for (int b = 0; b < nt; ++b) { iss += (h[b*3+1] - h[b*3]) / 100.0; tot += (h[b*3+2] - h[b*3]) / 100.0; if (h[b*3] < tmin) tmin = h[b*3]; if (h[b*3+2] > tmax) tmax = h[b*3+2]; }
        if (rep == 2) printf("M=%ld N=%ld tiles=%d pat %d: issue %.2f us, issue+drain %.2f us per workgroup (mean), span %.2f us\n", M, N, nt, pat, iss / nt, tot / nt, (tmax - tmin) / 100.0);
      }
    }
    CK(hipFree(C)); CK(hipFree(st));
  }
  return 0;
}
