// Semantics probe for the saturating packed rank sum K1 wants to use:
//   v_pk_sub_i16 ... clamp       -- per-half signed saturation of a - b
//   v_dot2_i32_i16 ... clamp     -- a.lo*b.lo + a.hi*b.hi + c evaluated exactly, then saturated to i32
// Prints the number of inputs where the hardware differs from that model.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include <random>

__global__ void probe(const uint32_t *a, const uint32_t *b, const int32_t *c, uint32_t *d_sub, int32_t *d_dot, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t s; int32_t t;
    asm volatile("v_pk_sub_i16 %0, %1, %2 clamp" : "=v"(s) : "v"(a[i]), "v"(b[i]));
    asm volatile("v_dot2_i32_i16 %0, %1, %2, %3 clamp\n s_nop 4" : "=v"(t) : "v"(a[i]), "v"(b[i]), "v"(c[i]));
    d_sub[i] = s; d_dot[i] = t;
}

static int sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }

int main() {
    std::vector<uint32_t> a, b; std::vector<int32_t> c;
    const int edge[] = {-32768, -32767, -1, 0, 1, 2, 32766, 32767, -16384, 16384, 23170, -23171};
    const int32_t cedge[] = {0, 1, -1, INT32_MAX, INT32_MIN, INT32_MAX - 1, 1 << 30, (1 << 30) + 65535, 0x3fff0001, INT32_MAX - (1 << 30)};
    for (int x : edge) for (int y : edge) for (int z : edge) for (int w : edge) for (int32_t k : cedge) {
        a.push_back((uint32_t)(x & 0xffff) | ((uint32_t)(y & 0xffff) << 16));
        b.push_back((uint32_t)(z & 0xffff) | ((uint32_t)(w & 0xffff) << 16));
        c.push_back(k);
    }
    std::mt19937 rng(7);
    for (int i = 0; i < 1 << 20; i++) { a.push_back(rng()); b.push_back(rng()); c.push_back((int32_t)rng()); }
    const int n = (int)a.size();
    uint32_t *da, *db, *ds; int32_t *dc, *dd;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
    probe<<<(n + 255) / 256, 256>>>(da, db, dc, ds, dd, n);
    std::vector<uint32_t> hs(n); std::vector<int32_t> hd(n);
    hipMemcpy(hs.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), dd, n * 4, hipMemcpyDeviceToHost);
    long bad_sub = 0, bad_dot = 0;
    for (int i = 0; i < n; i++) {
        int al = (int16_t)a[i], ah = (int16_t)(a[i] >> 16), bl = (int16_t)b[i], bh = (int16_t)(b[i] >> 16);
        uint32_t want_s = (uint32_t)(sat16(al - bl) & 0xffff) | ((uint32_t)(sat16(ah - bh) & 0xffff) << 16);
        long long t = (long long)al * bl + (long long)ah * bh + c[i];
        int32_t want_d = t > INT32_MAX ? INT32_MAX : t < INT32_MIN ? INT32_MIN : (int32_t)t;
        if (hs[i] != want_s) { if (bad_sub++ < 5) printf("pk_sub a=%08x b=%08x got %08x want %08x\n", a[i], b[i], hs[i], want_s); }
        if (hd[i] != want_d) { if (bad_dot++ < 5) printf("dot2 a=%08x b=%08x c=%d got %d want %d\n", a[i], b[i], c[i], hd[i], want_d); }
    }
    printf("clampsem: %d inputs, v_pk_sub_i16 clamp mismatches %ld, v_dot2_i32_i16 clamp mismatches %ld\n", n, bad_sub, bad_dot);
    return bad_sub || bad_dot ? 1 : 0;
}
