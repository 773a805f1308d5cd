// Scratch: host cost of a chain of N tiny dependent kernels, launched one by one vs as one hipGraph
// (stream capture), with 1 and 4 host threads each driving its own stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/ubench/graph_launch.hip -o /tmp/graph_launch -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(float *p, int i) { if (threadIdx.x == 0) p[i & 63] += 1.f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void chain(hipStream_t s, float *p, int n) { for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, p, i); }
int main() {
  const int N = 400, REP = 20;
  for (int threads : {1, 4}) {
    std::vector<double> t_plain(threads), t_graph(threads);
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t) th.emplace_back([&, t] {
      hipStream_t s; hipStreamCreate(&s);
      float *p; hipMalloc(&p, 256);
      chain(s, p, N); hipStreamSynchronize(s);
      double t0 = now();
      for (int r = 0; r < REP; ++r) chain(s, p, N);
      hipStreamSynchronize(s);
      t_plain[t] = (now() - t0) / REP;
      hipGraph_t g; hipGraphExec_t e;
      hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
      chain(s, p, N);
      hipStreamEndCapture(s, &g);
      hipGraphInstantiate(&e, g, nullptr, nullptr, 0);
      hipGraphLaunch(e, s); hipStreamSynchronize(s);
      t0 = now();
      for (int r = 0; r < REP; ++r) hipGraphLaunch(e, s);
      hipStreamSynchronize(s);
      t_graph[t] = (now() - t0) / REP;
    });
    for (auto &x : th) x.join();
    printf("%d thread(s): chain of %d kernels: plain %.0f us, graph %.0f us (thread 0)\n", threads, N, t_plain[0] * 1e6, t_graph[0] * 1e6);
  }
  return 0;
}
