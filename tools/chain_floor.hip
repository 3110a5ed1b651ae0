// What does a chain of DEPENDENT launches cost on this chip when the kernels do (almost) nothing?
// The decode step is 7 dependent launches (tools/README.md: profiles/r02_decode_step_launch_floor.txt).  Each kernel of
// the chain reads what the previous one wrote -- through a different workgroup, so the value crosses XCDs -- and
// writes `kb` KiB; 357 launches (51 steps x 7) from one host thread, eager and replayed from a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 tools/chain_floor.hip -o /tmp/chain_floor && /tmp/chain_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void link(const float* __restrict__ in, float* __restrict__ out, int n, int shift, int spin) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        int j = i + shift;               // a value another workgroup (other XCD: workgroups are dealt round-robin) wrote
        if (j >= n) j -= n;
        float v = in[j];
        for (int k = 0; k < spin; ++k) v = fmaf(v, 1.0001f, 1.0f);     // a dependent chain: `spin` x 4 cycles of body
        out[i] = v * 1.0001f + 1.0f;
    }
}

static int bench(int blocks, int spin, bool graph) {
    const int n = blocks * 256;
    float *a = nullptr, *b = nullptr;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int L = 357, reps = 20;
    auto enqueue = [&]() { for (int i = 0; i < L; ++i) hipLaunchKernelGGL(link, dim3(blocks), dim3(256), 0, st, (i & 1) ? b : a, (i & 1) ? a : b, n, 256 * 9 + 1, spin); };
    float ms = 0.f;
    if (!graph) {
        enqueue(); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) enqueue();
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
    } else {
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        enqueue();
        CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st)); CK(hipStreamSynchronize(st));
        CK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    printf("%-6s %5d workgroups (%6.0f KiB written per kernel), body spin %5d: %.2f us per dependent launch -> 7 launches = %.1f us, 5 = %.1f us\n",
           graph ? "graph" : "eager", blocks, n * 4 / 1024.0, spin, ms * 1e3 / (L * reps), 7 * ms * 1e3 / (L * reps), 5 * ms * 1e3 / (L * reps));
    hipFree(a); hipFree(b); hipStreamDestroy(st);
    return 0;
}

int main() {
    for (int graph = 0; graph < 2; ++graph)
        for (int blocks : {64, 256, 512, 1024, 4096}) if (bench(blocks, 0, graph != 0)) return 1;
    // with a body of a few microseconds the host is no longer the limit of the eager chain: is the gap the same?
    for (int spin : {500, 1000, 2000})
        for (int graph = 0; graph < 2; ++graph) if (bench(256, spin, graph != 0)) return 1;
    return 0;
}
