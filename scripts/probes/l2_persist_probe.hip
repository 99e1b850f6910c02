// l2_persist_probe.hip — does a line brought into an XCD's L2 by kernel N survive the kernel boundary, so that kernel N+1
// (same stream, inside a replayed hipGraph) hits it?  VERDICT r4 item 1 (prefetch the NEXT launch's weight panels from
// the current launch) rests on that.  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o l2probe l2_persist_probe.hip
//
// Workload shape = one 256 x 1024 x 1024 fp32 layer of the learner: a 4-MiB weight panel W, read by 256 workgroups whose
// XCD (blockIdx % 8) owns the 16-row blocks {x, x+8, ...} of W — tile_of_problem's map — i.e. 512 KiB per XCD.
//   reader<V>:  every workgroup reads ITS XCD's whole slice with 16-B loads (what the 8 q-tiles sharing a weight panel do
//               between them), folds it into one float per block.  Timed (kernel trace + events).
//   toucher:    one dword per 128-B line of the slice the SAME XCD will read next (64 lines per wave instruction).
// Sequences, each captured as one graph of 2 x NBUF kernels and replayed:
//   cold:     toucher(dummy) -> reader<0>(W[i])          W[i] rotates over NBUF panels (NBUF x 4 MiB > 32 MiB of L2)
//   warm:     toucher(W[i])  -> reader<1>(W[i])          prefetched by the previous kernel, same XCD map
//   crossed:  toucher(W[i], XCD+1) -> reader<2>(W[i])    prefetched into the WRONG XCD's L2 (control: must look cold)
//   self:     reader<3>(W[i]) -> reader<3>(W[i]) …       the same panel twice in a row (a reader after a reader)
// Output: average reader duration per sequence (HIP events around the graph, minus the toucher-only graph) — and, run under
// rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum (or FETCH_SIZE), the per-kernel counters by reader<V>.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kRows = 1024, kLd = 1024;            // W[1024][1024] fp32 = 4 MiB
constexpr int kBlocks = 256;

// rows of XCD x: 16-row blocks rb with rb % 8 == x (64 blocks of 16 rows -> 8 per XCD -> 128 rows = 512 KiB)
__device__ __forceinline__ const float* slice_row(const float* W, int xcd, int r) {   // r in [0, 128)
  const int rb = (r >> 4) * 8 + xcd;
  return W + (size_t)(rb * 16 + (r & 15)) * kLd;
}

template <int V>
__global__ __launch_bounds__(256) void reader(const float* __restrict__ W, float* __restrict__ out) {
  const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;      // 32 workgroups per XCD
  // each workgroup reads the whole slice (as the q-tiles of a GEMM do between them), starting at its own quarter
  float acc = 0.f;
  const int t = threadIdx.x;
  for (int rr = 0; rr < 128; ++rr) {
    const int r = (rr + j * 4) & 127;
    const float4 v = *reinterpret_cast<const float4*>(slice_row(W, xcd, r) + t * 4);
    acc += v.x + v.y + v.z + v.w;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;             // keep the loads
}

__global__ __launch_bounds__(256) void toucher(const float* __restrict__ W, float* __restrict__ out, int shift) {
  const int xcd = ((blockIdx.x & 7) + shift) & 7, j = blockIdx.x >> 3;
  // the XCD's slice = 128 rows x 4 KiB = 4096 lines; 32 workgroups x 256 threads = 8192 threads: one line per thread of
  // the first two waves of ... simply: thread g of the XCD (0..8191) touches line g if g < 4096
  const int g = j * 256 + threadIdx.x;
  float v = 0.f;
  if (g < 4096) v = slice_row(W, xcd, g >> 5)[(g & 31) * 32];
  if (v == 12345.678f) out[blockIdx.x] = v;
}

int main(int argc, char** argv) {
  const int NBUF = argc > 1 ? atoi(argv[1]) : 24, REPS = argc > 2 ? atoi(argv[2]) : 200;
  std::vector<float*> W(NBUF);
  float *dummy, *out;
  for (auto& p : W) { CK(hipMalloc(&p, (size_t)kRows * kLd * 4)); CK(hipMemset(p, 0, (size_t)kRows * kLd * 4)); }
  CK(hipMalloc(&dummy, (size_t)kRows * kLd * 4)); CK(hipMemset(dummy, 0, (size_t)kRows * kLd * 4));
  CK(hipMalloc(&out, 4096));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto run = [&](const char* name, auto enqueue) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
    enqueue();
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int i = 0; i < 5; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < REPS; ++i) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / REPS / NBUF;
    printf("%-10s %8.3f us per (kernel pair)\n", name, us);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return us;
  };
  const double t_touch = run("touch2", [&] { for (int i = 0; i < NBUF; ++i) { toucher<<<kBlocks, 256, 0, st>>>(dummy, out, 0); toucher<<<kBlocks, 256, 0, st>>>(dummy, out, 0); } });
  const double t_cold = run("cold", [&] { for (int i = 0; i < NBUF; ++i) { toucher<<<kBlocks, 256, 0, st>>>(dummy, out, 0); reader<0><<<kBlocks, 256, 0, st>>>(W[i], out); } });
  const double t_warm = run("warm", [&] { for (int i = 0; i < NBUF; ++i) { toucher<<<kBlocks, 256, 0, st>>>(W[i], out, 0); reader<1><<<kBlocks, 256, 0, st>>>(W[i], out); } });
  const double t_cross = run("crossed", [&] { for (int i = 0; i < NBUF; ++i) { toucher<<<kBlocks, 256, 0, st>>>(W[i], out, 1); reader<2><<<kBlocks, 256, 0, st>>>(W[i], out); } });
  const double t_self = run("self", [&] { for (int i = 0; i < NBUF; ++i) { reader<3><<<kBlocks, 256, 0, st>>>(W[i], out); reader<3><<<kBlocks, 256, 0, st>>>(W[i], out); } });
  printf("reader after an unrelated toucher (cold)      : %.3f us\n", t_cold - t_touch / 2);
  printf("reader after the toucher of ITS panel (warm)  : %.3f us   (toucher of a cold panel instead of a warm dummy included)\n", t_warm - t_touch / 2);
  printf("reader after a toucher on the wrong XCD       : %.3f us\n", t_cross - t_touch / 2);
  printf("reader after a reader of the same panel (avg) : %.3f us\n", t_self / 2);
  return 0;
}
