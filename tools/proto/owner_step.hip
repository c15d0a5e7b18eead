// tools/proto/owner_step.hip -- timing prototype (round 2): a two-kernel step for mid-size batches without a single
// item-side atomic.  A: per triplet, reads P[u], Q[i], Q[j]; writes P[u] (updated) and S[b] = g * P_pre[u].
// B: per distinct item, sums +-S[b] over the item's incidences (lists sorted by item, built on the host here)
// and updates Q[item] in place -- no G, no apply sweep.   hipcc -O3 --offload-arch=gfx950 owner_step.hip -o owner_step
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <math.h>
#include <vector>
#include <algorithm>
#include <numeric>
#include <string.h>
#include <rocprim/device/device_radix_sort.hpp>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
constexpr int D = 128, EPL = 4;
#ifndef CH
#define CH 8
#endif

__device__ __forceinline__ float gsum(float x) { for (int m = 16; m >= 1; m >>= 1) x += __shfl_xor(x, m, 64); return x; }

__global__ __launch_bounds__(256) void kernel_a(float* P, const float* Q, float* S, const int* U, const int* I, const int* J, int B, float lr, float invb)
{
    const int lane = threadIdx.x & 63, sub = lane >> 5, k = lane & 31;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int b = wave * 2 + sub; b < B; b += nw * 2) {
        const int u = U[b], i = I[b], j = J[b];
        float p[EPL], qi[EPL], qj[EPL];
        for (int c = 0; c < EPL; ++c) { p[c] = __builtin_nontemporal_load(P + (size_t)u * D + k + 32 * c); qi[c] = Q[(size_t)i * D + k + 32 * c]; qj[c] = Q[(size_t)j * D + k + 32 * c]; }
        float x = 0.f;
        for (int c = 0; c < EPL; ++c) x = fmaf(p[c], qi[c] - qj[c], x);
        x = gsum(x);
        const float g = -1.0f / (1.0f + __expf(x)) * invb;
        for (int c = 0; c < EPL; ++c) {
            __builtin_nontemporal_store(g * p[c], S + (size_t)b * D + k + 32 * c);
            __builtin_nontemporal_store(fmaf(-lr * g, qi[c] - qj[c], p[c]), P + (size_t)u * D + k + 32 * c);
        }
    }
}

// one lane group per SEGMENT (<= seg_len incidences of one item); inc[e] = b (positive) or ~b (negative)
__global__ __launch_bounds__(256) void kernel_b(float* Q, const float* S, const int* seg_item, const int* seg_begin, const int* seg_whole, const int* inc, int nseg, float lr)
{
    const int lane = threadIdx.x & 63, sub = lane >> 5, k = lane & 31;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    for (int s = wave * 2 + sub; s < nseg; s += nw * 2) {
        const int e0 = seg_begin[s], e1 = seg_begin[s + 1], item = seg_item[s];
        float acc[EPL] = {0.f, 0.f, 0.f, 0.f};
        int e = e0;
        for (; e + 1 < e1; e += 2) {                 // two rows in flight
            const int a = inc[e], bb = inc[e + 1];
            const float sa = a >= 0 ? 1.f : -1.f, sb = bb >= 0 ? 1.f : -1.f;
            const float* ra = S + (size_t)(a >= 0 ? a : ~a) * D + k;
            const float* rb = S + (size_t)(bb >= 0 ? bb : ~bb) * D + k;
            float va[EPL], vb[EPL];
            for (int c = 0; c < EPL; ++c) { va[c] = ra[32 * c]; vb[c] = rb[32 * c]; }
            for (int c = 0; c < EPL; ++c) acc[c] += sa * va[c] + sb * vb[c];
        }
        if (e < e1) {
            const int a = inc[e];
            const float sa = a >= 0 ? 1.f : -1.f;
            const float* ra = S + (size_t)(a >= 0 ? a : ~a) * D + k;
            for (int c = 0; c < EPL; ++c) acc[c] += sa * ra[32 * c];
        }
        float* q = Q + (size_t)item * D + k;
        if (seg_whole[s]) { for (int c = 0; c < EPL; ++c) q[32 * c] = fmaf(-lr, acc[c], q[32 * c]); }
        else { for (int c = 0; c < EPL; ++c) atomicAdd(q + 32 * c, -lr * acc[c]); }     // a popular item cut into several segments
    }
}


// B2: no segment arrays.  A lane group takes C consecutive incidences of the item-sorted list (keys = item, vals = b or ~b);
// runs of equal items that lie wholly inside the chunk are owned exclusively (plain read-modify-write of Q), runs that touch a
// chunk border go through atomics.
template <int C>
__global__ __launch_bounds__(256) void kernel_b2(float* Q, const float* S, const unsigned* keys, const int* vals, int n, float lr)
{
    const int lane = threadIdx.x & 63, sub = lane >> 5, k = lane & 31;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4;
    const int nchunk = (n + C - 1) / C;
    for (int ch = wave * 2 + sub; ch - sub < nchunk; ch += nw * 2) {
        const bool live = ch < nchunk;
        const int e0 = ch * C;
        // lane k of the group holds incidence e0 + k (C <= 32); lanes C, C+1 hold the neighbours outside the chunk
        unsigned mykey = 0xFFFFFFFFu; int myval = 0;
        if (live && k < C && e0 + k < n) { mykey = keys[e0 + k]; myval = vals[e0 + k]; }
        unsigned before = 0xFFFFFFFEu, after = 0xFFFFFFFDu;
        if (live && e0 > 0) before = keys[e0 - 1];
        if (live && e0 + C < n) after = keys[e0 + C];
        float acc[EPL] = {0.f, 0.f, 0.f, 0.f};
        unsigned run = 0xFFFFFFFFu; bool run_open_left = false;
        auto flush = [&](unsigned item, bool owned) {
            if (item == 0xFFFFFFFFu) return;
            float* q = Q + (size_t)item * D + k;
            if (owned) { for (int c = 0; c < EPL; ++c) q[32 * c] = fmaf(-lr, acc[c], q[32 * c]); }
            else { for (int c = 0; c < EPL; ++c) atomicAdd(q + 32 * c, -lr * acc[c]); }
        };
#pragma unroll
        for (int t0 = 0; t0 < C; t0 += 4) {
            unsigned it[4]; int vl[4]; float v[4][EPL];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                it[r] = __shfl(mykey, sub * 32 + t0 + r, 64); vl[r] = __shfl(myval, sub * 32 + t0 + r, 64);
                const float* row = S + (size_t)(vl[r] >= 0 ? vl[r] : ~vl[r]) * D + k;
                for (int c = 0; c < EPL; ++c) v[r][c] = (it[r] != 0xFFFFFFFFu) ? row[32 * c] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (it[r] == 0xFFFFFFFFu) continue;
                if (it[r] != run) {
                    flush(run, !run_open_left);               // a run that ended inside the chunk: owned unless it began before it
                    run = it[r]; run_open_left = (t0 + r == 0) && (before == run);
                    for (int c = 0; c < EPL; ++c) acc[c] = 0.f;
                }
                const float sg = vl[r] >= 0 ? 1.f : -1.f;
                for (int c = 0; c < EPL; ++c) acc[c] = fmaf(sg, v[r][c], acc[c]);
            }
        }
        flush(run, !run_open_left && after != run);
    }
}

int main(int argc, char** argv)
{
    const int U = 1000000, I = 100000;
    const int B = argc > 1 ? atoi(argv[1]) : 65536, seg_len = argc > 2 ? atoi(argv[2]) : 16;
    float *P, *Q, *S; int *dU, *dI, *dJ;
    CK(hipMalloc(&P, (size_t)U * D * 4)); CK(hipMalloc(&Q, (size_t)I * D * 4)); CK(hipMalloc(&S, (size_t)B * D * 4));
    std::vector<float> h((size_t)I * D); uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& x : h) x = ((rnd() % 2001) / 1000.0f - 1.0f) * 0.1f;
    CK(hipMemcpy(Q, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    for (int r = 0; r < U / I; ++r) CK(hipMemcpy(P + (size_t)r * I * D, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    // triplets: unique users (a random permutation prefix), Zipf-ish positives (item = I * u^3), uniform negatives
    std::vector<int> hu(U), hi(B), hj(B);
    std::iota(hu.begin(), hu.end(), 0);
    for (int q = U - 1; q > 0; --q) std::swap(hu[q], hu[rnd() % (q + 1)]);
    for (int b = 0; b < B; ++b) { double x = (rnd() % 1000000) / 1e6; hi[b] = (int)(I * x * x * x) % I; hj[b] = rnd() % I; }
    CK(hipMalloc(&dU, B * 4)); CK(hipMalloc(&dI, B * 4)); CK(hipMalloc(&dJ, B * 4));
    CK(hipMemcpy(dU, hu.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dI, hi.data(), B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dJ, hj.data(), B * 4, hipMemcpyHostToDevice));
    // incidence lists sorted by item, cut into segments of <= seg_len
    std::vector<std::pair<int, int>> incs;
    for (int b = 0; b < B; ++b) { incs.push_back({hi[b], b}); incs.push_back({hj[b], ~b}); }
    std::sort(incs.begin(), incs.end(), [](auto& a, auto& b) { return a.first < b.first; });
    std::vector<int> inc(2 * B), seg_item, seg_begin, seg_whole;
    int distinct = 0;
    for (int e = 0; e < 2 * B;) {
        int f = e; while (f < 2 * B && incs[f].first == incs[e].first) ++f;
        ++distinct;
        const bool whole = f - e <= seg_len;
        for (int g = e; g < f; g += seg_len) { seg_item.push_back(incs[e].first); seg_begin.push_back(g); seg_whole.push_back(whole); }
        e = f;
    }
    seg_begin.push_back(2 * B);
    for (int e = 0; e < 2 * B; ++e) inc[e] = incs[e].second;
    const int nseg = (int)seg_item.size();
    int *dinc, *dsi, *dsb, *dsw;
    CK(hipMalloc(&dinc, 2 * B * 4)); CK(hipMalloc(&dsi, nseg * 4)); CK(hipMalloc(&dsb, (nseg + 1) * 4)); CK(hipMalloc(&dsw, nseg * 4));
    CK(hipMemcpy(dinc, inc.data(), 2 * B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsi, seg_item.data(), nseg * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, seg_begin.data(), (nseg + 1) * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsw, seg_whole.data(), nseg * 4, hipMemcpyHostToDevice));
    printf("B=%d distinct items %d, segments %d (<= %d incidences)\n", B, distinct, nseg, seg_len);
    hipEvent_t e0, e1, e2; hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
    const int ga = std::min(2048, (B / 2 + 3) / 4), gb = std::min(2048, (nseg / 2 + 3) / 4);
    float ta = 0, tb = 0, tt = 0;
    for (int it = 0; it < 13; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(kernel_a, dim3(ga), dim3(256), 0, 0, P, Q, S, dU, dI, dJ, B, 0.05f, 1.0f / B);
        hipEventRecord(e1);
        hipLaunchKernelGGL(kernel_b, dim3(gb), dim3(256), 0, 0, Q, S, dsi, dsb, dsw, dinc, nseg, 0.05f);
        hipEventRecord(e2); hipEventSynchronize(e2);
        float a, b, t; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2); hipEventElapsedTime(&t, e0, e2);
        if (it >= 3) { ta += a; tb += b; tt += t; }
    }
    printf("kernel A (users + S rows) %.1f us, kernel B (item sums + in-place Q) %.1f us, both %.1f us\n", ta * 100, tb * 100, tt * 100);
    // device side: sort of the 2B (item, +-b) incidences with rocPRIM, then the chunked kernel
    unsigned *k_in, *k_out; int *v_in, *v_out;
    CK(hipMalloc(&k_in, 2 * B * 4)); CK(hipMalloc(&k_out, 2 * B * 4)); CK(hipMalloc(&v_in, 2 * B * 4)); CK(hipMalloc(&v_out, 2 * B * 4));
    std::vector<unsigned> hk(2 * B); std::vector<int> hv(2 * B);
    for (int b = 0; b < B; ++b) { hk[b] = hi[b]; hv[b] = b; hk[B + b] = hj[b]; hv[B + b] = ~b; }
    CK(hipMemcpy(k_in, hk.data(), 2 * B * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(v_in, hv.data(), 2 * B * 4, hipMemcpyHostToDevice));
    size_t tmp_bytes = 0; void* tmp = nullptr;
    CK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)(2 * B), 0, 17, 0));
    CK(hipMalloc(&tmp, tmp_bytes));
    float tsort = 0, tb2 = 0;
    for (int it = 0; it < 13; ++it) {
        hipEventRecord(e0);
        CK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k_in, k_out, v_in, v_out, (size_t)(2 * B), 0, 17, 0));
        hipEventRecord(e1);
        const int nchunk = (2 * B + CH - 1) / CH;
        hipLaunchKernelGGL(kernel_b2<CH>, dim3(std::min(2048, (nchunk / 2 + 3) / 4 + 1)), dim3(256), 0, 0, Q, S, k_out, v_out, 2 * B, 0.05f);
        hipEventRecord(e2); hipEventSynchronize(e2);
        float a, b; hipEventElapsedTime(&a, e0, e1); hipEventElapsedTime(&b, e1, e2);
        if (it >= 3) { tsort += a; tb2 += b; }
    }
    printf("rocPRIM radix sort of %d pairs (17 bits) %.1f us (temp %zu B); chunked kernel B2 (C = %d) %.1f us\n", 2 * B, tsort * 100, tmp_bytes, CH, tb2 * 100);
    return 0;
}
