// tools/row_width_bench.hip -- development probe (MI355X): the memory skeleton of the BPR step kernel -- per triplet a random user row
// read and written back (streaming), two item rows read from a 100K-row table -- with the two candidate register layouts of a row:
//   W = 1  lane k of a 32-lane group holds elements k, k+32, ...   (D/32 global_load_dword per row: one 128-B line per row per instruction;
//          the layout the fp32 atomics want, csrc/rsx_bpr.hip "ROW LAYOUT")
//   W = 4  lane k of a D/4-lane group holds elements 4k .. 4k+3     (ONE global_load_dwordx4 per row; 64 / (D/4) rows per wave instruction)
// Same positions per lane group, same two positions in flight per trip.  What it answers: is the step kernel bound by the number of
// vector-memory instructions (address processing of dword accesses) rather than by bytes?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <vector>
#include <utility>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int D, int W> struct Lay;
template <int D> struct Lay<D, 1> { static constexpr int LPR = 32, EPL = D / 32; };
template <int D> struct Lay<D, 4> { static constexpr int LPR = D / 4, EPL = 4; };

template <int D, int W>
__device__ __forceinline__ void load_row(const float *base, int row, int k, float (&v)[Lay<D, W>::EPL], bool nt)
{
    if constexpr (W == 1) {
#pragma unroll
        for (int c = 0; c < D / 32; ++c) v[c] = nt ? __builtin_nontemporal_load(base + (size_t)row * D + k + 32 * c) : base[(size_t)row * D + k + 32 * c];
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *p = reinterpret_cast<const f4 *>(base + (size_t)row * D) + k;
        const f4 t = nt ? __builtin_nontemporal_load(p) : *p;
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
}
template <int D, int W>
__device__ __forceinline__ void store_row(float *base, int row, int k, const float (&v)[Lay<D, W>::EPL])
{
    if constexpr (W == 1) {
#pragma unroll
        for (int c = 0; c < D / 32; ++c) __builtin_nontemporal_store(v[c], base + (size_t)row * D + k + 32 * c);
    } else {
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 t = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(t, reinterpret_cast<f4 *>(base + (size_t)row * D) + k);
    }
}

// MODE 0: read P, Q[i], Q[j], write P   1: P read + write only   2: the three reads only
template <int D, int W, int MODE>
__global__ __launch_bounds__(256, 6) void skel(float *P, const float *Q, const int *U, const int *I, const int *J, int64_t B, int per_wave)
{
    constexpr int LPR = Lay<D, W>::LPR, EPL = Lay<D, W>::EPL, TPW = 64 / LPR;
    const int lane = threadIdx.x & 63, sub = lane / LPR, k = lane % LPR;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t b0 = wave * per_wave, b1 = (b0 + per_wave < B) ? b0 + per_wave : B;
    if (b0 >= B) return;
    const int64_t len = (b1 - b0 + TPW - 1) / TPW, g_lo = b0 + sub * len, g_hi = (g_lo + len < b1) ? g_lo + len : b1;
    float acc = 0.f;
    for (int64_t t = 0; t < len; t += 2) {
        const int64_t pa = g_lo + t, pb = g_lo + t + 1;
        const bool la = pa < g_hi, lb = pb < g_hi;
        int ua = 0, ia = 0, ja = 0, ub = 0, ib = 0, jb = 0;
        if (la) { ua = U[pa]; ia = I[pa]; ja = J[pa]; }
        if (lb) { ub = U[pb]; ib = I[pb]; jb = J[pb]; }
        float xa[EPL], ya[EPL], za[EPL], xb[EPL], yb[EPL], zb[EPL];
#pragma unroll
        for (int c = 0; c < EPL; ++c) { xa[c] = ya[c] = za[c] = xb[c] = yb[c] = zb[c] = 0.f; }
        if (la) { load_row<D, W>(P, ua, k, xa, true); if (MODE != 1) { load_row<D, W>(Q, ia, k, ya, false); load_row<D, W>(Q, ja, k, za, false); } }
        if (lb) { load_row<D, W>(P, ub, k, xb, true); if (MODE != 1) { load_row<D, W>(Q, ib, k, yb, false); load_row<D, W>(Q, jb, k, zb, false); } }
#pragma unroll
        for (int c = 0; c < EPL; ++c) { acc += xa[c] * (ya[c] - za[c]) + xb[c] * (yb[c] - zb[c]); xa[c] += 1e-3f * (ya[c] - za[c]); xb[c] += 1e-3f * (yb[c] - zb[c]); }
        if (MODE != 2) { if (la) store_row<D, W>(P, ua, k, xa); if (lb) store_row<D, W>(P, ub, k, xb); }
    }
    if (acc == 123.456f) P[0] = acc;
}

int main(int argc, char **argv)
{
    const int64_t Un = argc > 1 ? atoll(argv[1]) : 1000000, In = argc > 2 ? atoll(argv[2]) : 100000, B = Un;
    const int per_wave = argc > 3 ? atoi(argv[3]) : 20;
    float *P, *Q; int *U, *I, *J;
    CK(hipMalloc(&P, Un * 128 * 4)); CK(hipMemset(P, 0, Un * 128 * 4));
    CK(hipMalloc(&Q, In * 128 * 4)); CK(hipMemset(Q, 0, In * 128 * 4));
    std::vector<int> hu(B), hi(B), hj(B); uint64_t s = 88172645463325252ull;
    for (int64_t q = 0; q < B; ++q) hu[q] = (int)q;
    for (int64_t q = B - 1; q > 0; --q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(hu[q], hu[s % (q + 1)]); }
    for (int64_t q = 0; q < B; ++q) { hi[q] = (int)((double)q / B * In); s ^= s << 13; s ^= s >> 7; s ^= s << 17; hj[q] = (int)(s % In); }   // positives ordered, negatives random
    CK(hipMalloc(&U, B * 4)); CK(hipMalloc(&I, B * 4)); CK(hipMalloc(&J, B * 4));
    CK(hipMemcpy(U, hu.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(I, hi.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(J, hj.data(), B * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned blocks = (unsigned)(((B + per_wave - 1) / per_wave + 3) / 4);
    printf("users %lld items %lld triplets %lld, %d positions per wavefront\n", (long long)Un, (long long)In, (long long)B, per_wave);
#define RUN(D, W, M, name) { for (int it = 0; it < 3; ++it) hipLaunchKernelGGL((skel<D, W, M>), dim3(blocks), dim3(256), 0, 0, P, Q, U, I, J, B, per_wave); \
    hipEventRecord(e0); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL((skel<D, W, M>), dim3(blocks), dim3(256), 0, 0, P, Q, U, I, J, B, per_wave); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; printf("d=%-3d %-8s %-28s %8.1f us\n", D, W == 1 ? "dword" : "dwordx4", name, ms * 1e3); }
    RUN(128, 1, 0, "P r/w + Q[i] + Q[j]") RUN(128, 4, 0, "P r/w + Q[i] + Q[j]")
    RUN(128, 1, 1, "P r/w only") RUN(128, 4, 1, "P r/w only")
    RUN(128, 1, 2, "reads only") RUN(128, 4, 2, "reads only")
    RUN(64, 1, 0, "P r/w + Q[i] + Q[j]") RUN(64, 4, 0, "P r/w + Q[i] + Q[j]")
    RUN(64, 1, 1, "P r/w only") RUN(64, 4, 1, "P r/w only")
    RUN(64, 1, 2, "reads only") RUN(64, 4, 2, "reads only")
    return 0;
}
