// Microbenchmark: per-kernel cost of a dependent chain of small kernels, eager vs hipGraph.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void empty_k(int *p) { }
__global__ void touch_k(float *p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1.f; }
struct Big { char pad[320]; float *p; int n; };
__global__ void bigarg_k(Big b) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < b.n) b.p[i] += 1.f; }
int main() {
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  float *d; CK(hipMalloc(&d, 1 << 24)); CK(hipMemset(d, 0, 1 << 24));
  const int N = 2000;
  auto run = [&](const char *name, auto launch) {
    for (int i = 0; i < 200; ++i) launch(st);
    hipStreamSynchronize(st);
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < N; ++i) launch(st);
    hipStreamSynchronize(st);
    double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("eager %-28s %.2f us/kernel\n", name, us / N);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < 100; ++i) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int i = 0; i < 5; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    printf("graph %-28s %.2f us/kernel (100-node graph, %.1f us/replay)\n", name, us / 2000, us / 20);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  };
  run("empty<<<1,64>>>", [&](hipStream_t s) { empty_k<<<1, 64, 0, s>>>(nullptr); });
  run("empty<<<256,256>>>", [&](hipStream_t s) { empty_k<<<256, 256, 0, s>>>(nullptr); });
  run("touch 64KB <<<64,256>>>", [&](hipStream_t s) { touch_k<<<64, 256, 0, s>>>(d, 16384); });
  run("touch 4MB <<<4096,256>>>", [&](hipStream_t s) { touch_k<<<4096, 256, 0, s>>>(d, 1 << 20); });
  Big b; b.p = d; b.n = 16384;
  run("bigarg 64KB <<<64,256>>>", [&](hipStream_t s) { bigarg_k<<<64, 256, 0, s>>>(b); });
  return 0;
}
