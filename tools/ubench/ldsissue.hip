// What does an LDS instruction cost a LONE wavefront in ISSUE time -- beyond its latency, which K1's slice loop hides?
// Loop body: 16 dependent-free VALU (v_add_u32 on 16 different registers; 4.07 cycles each for a lone wavefront) plus a set of
// LDS instructions whose results nobody waits for inside the loop (one s_waitcnt lgkmcnt(0) every 64 iterations keeps the
// queue bounded).  cycles per iteration - the VALU-only figure = what the LDS instructions add to the issue stream.
// K1's slice loop has, per slice: 2 x ds_write_b128 (candidate state, 32-byte slots), 2 x ds_read_b128 (winner's slot, row
// broadcast), 3 x ds_read_b128 (the next slice's samples, row broadcast), 1 x ds_write_b64 (the winner's slice word).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/ldsissue.hip -o tools/ubench/ldsissue
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 4096
#define VALU16                                                                                                              \
    "v_add_u32 v20, 1, v20\n v_add_u32 v21, 1, v21\n v_add_u32 v22, 1, v22\n v_add_u32 v23, 1, v23\n"                         \
    "v_add_u32 v24, 1, v24\n v_add_u32 v25, 1, v25\n v_add_u32 v26, 1, v26\n v_add_u32 v27, 1, v27\n"                         \
    "v_add_u32 v28, 1, v28\n v_add_u32 v29, 1, v29\n v_add_u32 v30, 1, v30\n v_add_u32 v31, 1, v31\n"                         \
    "v_add_u32 v32, 1, v32\n v_add_u32 v33, 1, v33\n v_add_u32 v34, 1, v34\n v_add_u32 v35, 1, v35\n"
#define CLOB "v10","v11","v12","v13","v14","v15","v16","v17","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","memory"
template <int MODE>
__global__ void k(unsigned long long *out, int *sink) {
    __shared__ __attribute__((aligned(512))) int lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u;
    unsigned slot32 = lane * 32;                    // K1's candidate slots: 32 bytes apart
    unsigned slot16 = lane * 16;                    // the same state as two conflict-free 16-byte arrays (lab idea)
    unsigned row = (lane & 0x30) * 32;              // one address per 16-lane row: broadcast reads
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < REP; i++) {
        if (MODE == 0) asm volatile(VALU16 ::: CLOB);
        else if (MODE == 1) asm volatile("ds_write_b128 %0, v[10:13]\n" VALU16 :: "v"(slot32) : CLOB);
        else if (MODE == 2) asm volatile("ds_write_b128 %0, v[10:13]\n ds_write_b128 %0, v[14:17] offset:16\n" VALU16 :: "v"(slot32) : CLOB);
        else if (MODE == 3) asm volatile("ds_write_b128 %0, v[10:13]\n ds_write_b128 %0, v[14:17] offset:4096\n" VALU16 :: "v"(slot16) : CLOB);
        else if (MODE == 4) asm volatile("ds_read_b128 v[40:43], %0\n" VALU16 :: "v"(row) : CLOB);
        else if (MODE == 5) asm volatile("ds_read_b128 v[40:43], %0\n ds_read_b128 v[44:47], %0 offset:16\n ds_read_b128 v[48:51], %0 offset:32\n" VALU16 :: "v"(row) : CLOB);
        else if (MODE == 6) asm volatile("ds_write_b64 %0, v[10:11]\n" VALU16 :: "v"(slot32) : CLOB);
        else if (MODE == 7) // the slice loop's whole LDS traffic, spread through the VALU as in the loop
            asm volatile("ds_write_b128 %0, v[10:13]\n ds_write_b128 %0, v[14:17] offset:16\n v_add_u32 v20, 1, v20\n v_add_u32 v21, 1, v21\n"
                         "v_add_u32 v22, 1, v22\n v_add_u32 v23, 1, v23\n v_add_u32 v24, 1, v24\n v_add_u32 v25, 1, v25\n v_add_u32 v26, 1, v26\n"
                         "v_add_u32 v27, 1, v27\n ds_read_b128 v[40:43], %1\n ds_read_b128 v[44:47], %1 offset:16\n v_add_u32 v28, 1, v28\n"
                         "v_add_u32 v29, 1, v29\n ds_write_b64 %0, v[10:11] offset:2048\n v_add_u32 v30, 1, v30\n v_add_u32 v31, 1, v31\n"
                         "ds_read_b128 v[48:51], %1 offset:32\n v_add_u32 v32, 1, v32\n ds_read_b128 v[48:51], %1 offset:48\n v_add_u32 v33, 1, v33\n"
                         "ds_read_b128 v[48:51], %1 offset:64\n v_add_u32 v34, 1, v34\n v_add_u32 v35, 1, v35\n" :: "v"(slot32), "v"(row) : CLOB);
        else if (MODE == 8) // ... the same with the candidate state in two conflict-free 16-byte arrays
            asm volatile("ds_write_b128 %0, v[10:13]\n ds_write_b128 %0, v[14:17] offset:4096\n v_add_u32 v20, 1, v20\n v_add_u32 v21, 1, v21\n"
                         "v_add_u32 v22, 1, v22\n v_add_u32 v23, 1, v23\n v_add_u32 v24, 1, v24\n v_add_u32 v25, 1, v25\n v_add_u32 v26, 1, v26\n"
                         "v_add_u32 v27, 1, v27\n ds_read_b128 v[40:43], %1\n ds_read_b128 v[44:47], %1 offset:4096\n v_add_u32 v28, 1, v28\n"
                         "v_add_u32 v29, 1, v29\n ds_write_b64 %0, v[10:11] offset:2048\n v_add_u32 v30, 1, v30\n v_add_u32 v31, 1, v31\n"
                         "ds_read_b128 v[48:51], %1 offset:32\n v_add_u32 v32, 1, v32\n ds_read_b128 v[48:51], %1 offset:48\n v_add_u32 v33, 1, v33\n"
                         "ds_read_b128 v[48:51], %1 offset:64\n v_add_u32 v34, 1, v34\n v_add_u32 v35, 1, v35\n" :: "v"(slot16), "v"(row) : CLOB);
        else if (MODE == 9)  // the two state writes under an EXEC mask of 4 lanes (one winner per row)
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %1\n ds_write_b128 %0, v[10:13]\n ds_write_b128 %0, v[14:17] offset:16\n"
                         "s_mov_b64 exec, s[20:21]\n" VALU16 :: "v"(slot32), "s"(0x0001000100010001ull) : CLOB, "s20", "s21");
        else if (MODE == 10) // ds_write_b64 under the same mask
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %1\n ds_write_b64 %0, v[10:11]\n"
                         "s_mov_b64 exec, s[20:21]\n" VALU16 :: "v"(slot32), "s"(0x0001000100010001ull) : CLOB, "s20", "s21");
        else if (MODE == 11) // only the EXEC save / set / restore (3 SALU)
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, %1\n"
                         "s_mov_b64 exec, s[20:21]\n" VALU16 :: "v"(slot32), "s"(0x0001000100010001ull) : CLOB, "s20", "s21");
        else if (MODE == 12) // 2 x ds_write_b64 instead of one b128 (is it the width or the instruction?)
            asm volatile("ds_write_b64 %0, v[10:11]\n ds_write_b64 %0, v[12:13] offset:8\n" VALU16 :: "v"(slot32) : CLOB);
        else if (MODE == 13) // ds_write_b32
            asm volatile("ds_write_b32 %0, v10\n" VALU16 :: "v"(slot32) : CLOB);
        else if (MODE == 14) // ds_read_b128, every lane its own address (no broadcast)
            asm volatile("ds_read_b128 v[40:43], %0\n" VALU16 :: "v"(slot32) : CLOB);
        if ((i & 63) == 63) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0) out[threadIdx.x >> 6] = t1 - t0;
    sink[threadIdx.x] = (int)slot32;
}
template <int MODE> double run(const char *name, unsigned long long *out, int *sink, double base) {
    for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k<MODE>, dim3(1), dim3(64), 0, 0, out, sink);
    hipDeviceSynchronize();
    unsigned long long h; hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    const double c = (double)h / REP;
    printf("%-88s %7.1f cycles per iteration  (+%.1f over 16 VALU)\n", name, c, c - base);
    return c;
}
int main() {
    unsigned long long *out; int *sink; hipMalloc(&out, 64); hipMalloc(&sink, 4096);
    const double b = run<0>("16 x v_add_u32 (lone wavefront)", out, sink, 0);
    run<1>("+ 1 x ds_write_b128, 32-byte slots (2 lanes per bank group)", out, sink, b);
    run<2>("+ 2 x ds_write_b128, 32-byte slots: K1's candidate state", out, sink, b);
    run<3>("+ 2 x ds_write_b128, two 16-byte arrays (conflict-free)", out, sink, b);
    run<4>("+ 1 x ds_read_b128 row broadcast", out, sink, b);
    run<5>("+ 3 x ds_read_b128 row broadcast: the next slice's samples", out, sink, b);
    run<6>("+ 1 x ds_write_b64", out, sink, b);
    run<7>("+ the slice loop's 8 LDS instructions (2 w128, 2 r128, w64, 3 r128), spread as in the loop", out, sink, b);
    run<8>("+ the same with the candidate state in two conflict-free 16-byte arrays", out, sink, b);
    run<11>("+ EXEC save / set to 4 lanes / restore (3 SALU)", out, sink, b);
    run<9>("+ 2 x ds_write_b128 under an EXEC mask of 4 lanes (+ the 3 SALU)", out, sink, b);
    run<10>("+ 1 x ds_write_b64 under an EXEC mask of 4 lanes (+ the 3 SALU)", out, sink, b);
    run<12>("+ 2 x ds_write_b64 (all lanes)", out, sink, b);
    run<13>("+ 1 x ds_write_b32 (all lanes)", out, sink, b);
    run<14>("+ 1 x ds_read_b128, every lane its own 32-byte slot", out, sink, b);
    return 0;
}
