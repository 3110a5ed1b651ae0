// Experiment for the next round's design: can two host threads, each feeding its own stream, launch small
// dependent kernels at twice the rate of one thread?  (Needed for a dual-chain decode recurrence.)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>
__global__ void tiny(float* p, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}
static double run(int nthreads, int launches, int blocks) {
    std::vector<hipStream_t> st(nthreads);
    std::vector<float*> buf(nthreads);
    for (int t = 0; t < nthreads; ++t) { hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking); hipMalloc(&buf[t], blocks * 256 * 4); }
    hipDeviceSynchronize();
    auto t0 = std::chrono::high_resolution_clock::now();
    std::vector<std::thread> th;
    for (int t = 0; t < nthreads; ++t)
        th.emplace_back([&, t] {
            hipSetDevice(0);
            for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(256), 0, st[t], buf[t], blocks * 256);
        });
    for (auto& x : th) x.join();
    auto t1 = std::chrono::high_resolution_clock::now();
    hipDeviceSynchronize();
    auto t2 = std::chrono::high_resolution_clock::now();
    double host_us = std::chrono::duration<double, std::micro>(t1 - t0).count();
    double total_us = std::chrono::duration<double, std::micro>(t2 - t0).count();
    printf("threads %d blocks %4d: host enqueue %.2f us/launch/thread, end-to-end %.2f us per launch-slot (%.2f us per kernel overall)\n",
           nthreads, blocks, host_us / launches, total_us / launches, total_us / launches / nthreads);
    for (int t = 0; t < nthreads; ++t) { hipFree(buf[t]); hipStreamDestroy(st[t]); }
    return total_us;
}
int main() {
    for (int blocks : {64, 256, 1024}) {
        run(1, 20000, blocks);
        run(2, 20000, blocks);
        run(4, 20000, blocks);
    }
    return 0;
}
