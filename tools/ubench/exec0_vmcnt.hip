// Does a vector-memory instruction issued with EXEC = 0 take part in the vmcnt order?
// K2u's static vmcnt bookkeeping (qoa_decode.hip: K2uCfg::behind) counts store instructions that may execute with every
// lane masked off (the absolute-line variant's second store slot; the masked role group of 3 / 5 / 6 / 7 channels never has an
// empty mask).  If such a store were dropped WITHOUT passing through the counter, `s_waitcnt vmcnt(K)` behind [load, K x
// store(EXEC = 0)] would be satisfied while the load is still in flight.
// Test: one cold global load (a fresh cache line per iteration: ~1-2 us), K stores under EXEC = 0, s_waitcnt vmcnt(K), then
// read the clock.  Counted: the wait lasts a memory latency (as with K real stores).  Not counted: a few cycles.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/exec0_vmcnt.hip -o tools/ubench/exec0_vmcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
template <int MODE>
__global__ void k(const unsigned *src, unsigned *dst, unsigned long long *out, unsigned *sink) {
    const unsigned lane = threadIdx.x;
    unsigned long long acc = 0;
    unsigned got = 0;
    for (int i = 0; i < REP; i++) {
        const unsigned *p = src + ((size_t)i * 4099u + blockIdx.x * 977u) * 32u + lane; // a line nobody has touched
        unsigned *q = dst + (size_t)i * 64u + lane;
        unsigned v, t0, t1;
        if (MODE == 0)      // the load alone, waited for
            asm volatile("s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n global_load_dword %0, %3, off\n s_waitcnt vmcnt(0)\n s_memtime s[22:23]\n"
                         "s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, s20\n v_mov_b32 %2, s22\n"
                         : "=&v"(v), "=&v"(t0), "=&v"(t1) : "v"(p) : "s20", "s21", "s22", "s23", "memory");
        else if (MODE == 1) // load, 4 real stores, vmcnt(4): must still wait for the load
            asm volatile("s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n global_load_dword %0, %3, off\n"
                         "global_store_dword %4, %5, off\n global_store_dword %4, %5, off offset:256\n global_store_dword %4, %5, off offset:512\n"
                         "global_store_dword %4, %5, off offset:768\n s_waitcnt vmcnt(4)\n s_memtime s[22:23]\n"
                         "s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, s20\n v_mov_b32 %2, s22\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(v), "=&v"(t0), "=&v"(t1) : "v"(p), "v"(q), "v"(lane) : "s20", "s21", "s22", "s23", "memory");
        else if (MODE == 2) // load, 4 stores under EXEC = 0, vmcnt(4)
            asm volatile("s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n global_load_dword %0, %3, off\n s_mov_b64 s[24:25], exec\n s_mov_b64 exec, 0\n"
                         "global_store_dword %4, %5, off\n global_store_dword %4, %5, off offset:256\n global_store_dword %4, %5, off offset:512\n"
                         "global_store_dword %4, %5, off offset:768\n s_mov_b64 exec, s[24:25]\n s_waitcnt vmcnt(4)\n s_memtime s[22:23]\n"
                         "s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, s20\n v_mov_b32 %2, s22\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(v), "=&v"(t0), "=&v"(t1) : "v"(p), "v"(q), "v"(lane) : "s20", "s21", "s22", "s23", "s24", "s25", "memory");
        else                // load, nothing behind it, vmcnt(4): returns at once (the control for "not counted")
            asm volatile("s_memtime s[20:21]\n s_waitcnt lgkmcnt(0)\n global_load_dword %0, %3, off\n s_waitcnt vmcnt(4)\n s_memtime s[22:23]\n"
                         "s_waitcnt lgkmcnt(0)\n v_mov_b32 %1, s20\n v_mov_b32 %2, s22\n s_waitcnt vmcnt(0)\n"
                         : "=&v"(v), "=&v"(t0), "=&v"(t1) : "v"(p) : "s20", "s21", "s22", "s23", "memory");
        acc += (unsigned)(t1 - t0);
        got += v;
    }
    if (lane == 0) out[blockIdx.x] = acc;
    sink[blockIdx.x * 64 + lane] = got;
}
template <int MODE> void run(const char *name, const unsigned *src, unsigned *dst, unsigned long long *out, unsigned *sink) {
    hipLaunchKernelGGL(k<MODE>, dim3(8), dim3(64), 0, 0, src, dst, out, sink);
    (void)hipDeviceSynchronize();
    unsigned long long h[8]; (void)hipMemcpy(h, out, sizeof h, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    printf("%-70s %8.1f s_memtime ticks per iteration\n", name, s / 8 / REP);
}
int main() {
    unsigned *src, *dst, *sink; unsigned long long *out;
    (void)hipMalloc(&src, (size_t)64 * 4099 * 32 * 4 + (1 << 20)); (void)hipMalloc(&dst, 1 << 20); (void)hipMalloc(&out, 64); (void)hipMalloc(&sink, 8 * 64 * 4);
    (void)hipMemset(src, 1, (size_t)64 * 4099 * 32 * 4 + (1 << 20));
    run<0>("cold load + vmcnt(0)", src, dst, out, sink);
    run<1>("cold load, 4 stores, vmcnt(4)", src, dst, out, sink);
    run<2>("cold load, 4 stores under EXEC = 0, vmcnt(4)", src, dst, out, sink);
    run<3>("cold load, nothing behind it, vmcnt(4) (no wait)", src, dst, out, sink);
    return 0;
}
