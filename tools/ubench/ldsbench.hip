#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 256
template <int MODE>
__global__ void k(unsigned long long *out, int *sink) {
    __shared__ __attribute__((aligned(256))) int lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    unsigned addr = (threadIdx.x & 0xf0) * 16;   // one 256-B row per 16 lanes
    unsigned own = threadIdx.x * 16;
    int acc = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        if (MODE == 0)      // ds_read_b128 broadcast per row, dependent
            asm volatile("ds_read_b128 v[10:13], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","v11","v12","v13","memory");
        else if (MODE == 1) // ds_read_b32 broadcast, dependent
            asm volatile("ds_read_b32 v10, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","memory");
        else if (MODE == 2) // bpermute dependent
            asm volatile("ds_bpermute_b32 v10, %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","memory");
        else if (MODE == 3) // write b128 own slot + read b128 winner slot, dependent (the exchange)
            asm volatile("ds_write_b128 %1, v[10:13]\n ds_read_b128 v[10:13], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) : "v"(own) : "v10","v11","v12","v13","memory");
        else if (MODE == 4) // 5 sample reads ahead of the state read
            asm volatile("ds_read_b128 v[14:17], %1\n ds_read_b128 v[18:21], %1 offset:16\n ds_read_b128 v[22:25], %1 offset:32\n ds_read_b128 v[26:29], %1 offset:48\n ds_read_b128 v[30:33], %1 offset:64\n"
                         "ds_read_b128 v[10:13], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) : "v"(own) : "v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","memory");
        else if (MODE == 5) // ds_read_b64 broadcast dependent
            asm volatile("ds_read_b64 v[10:11], %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","v11","memory");
        else if (MODE == 6) // 2 x ds_read_b64 
            asm volatile("ds_read2_b64 v[10:13], %0 offset1:1\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","v11","v12","v13","memory");
        else if (MODE == 7) // 4 x bpermute pipelined then wait
            asm volatile("ds_bpermute_b32 v10, %0, %0\n ds_bpermute_b32 v11, %0, %0\n ds_bpermute_b32 v12, %0, %0\n ds_bpermute_b32 v13, %0, %0\n s_waitcnt lgkmcnt(0)\n v_and_b32 v10, 0xf0, v10\n v_or_b32 %0, v10, %0\n" : "+v"(addr) :: "v10","v11","v12","v13","memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = addr + acc;
}
template <int MODE> void run(const char *name, unsigned long long *out, int *sink) {
    for (int wv : {1, 4}) {
        for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64 * wv), 0, 0, out, sink);
        hipDeviceSynchronize();
        unsigned long long h[4]; hipMemcpy(h, out, 8 * wv, hipMemcpyDeviceToHost);
        printf("%-50s waves/CU=%d  %.1f cycles per iteration\n", name, wv, (double)h[0] / REP);
    }
}
int main() {
    unsigned long long *out; int *sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    run<0>("ds_read_b128 (row broadcast) + wait + 1 valu", out, sink);
    run<1>("ds_read_b32 (row broadcast) + wait + 1 valu", out, sink);
    run<5>("ds_read_b64 (row broadcast) + wait + 1 valu", out, sink);
    run<6>("ds_read2_b64 (row broadcast) + wait + 1 valu", out, sink);
    run<2>("ds_bpermute_b32 + wait + 1 valu", out, sink);
    run<7>("4 x ds_bpermute_b32 + wait + 1 valu", out, sink);
    run<3>("ds_write_b128 own + ds_read_b128 row + wait + 1", out, sink);
    run<4>("5 x ds_read_b128 + ds_read_b128 + wait + 1", out, sink);
    return 0;
}
