// tools/atomic_bench.hip -- development probe: device-scope atomic throughput on MI355X
// by operand type and access shape (random 512-B rows, 128 B per row per instruction).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <stdlib.h>
#include <utility>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

// each 32-lane half-wave owns one random row of 128 floats (512 B); MODE selects the op
template <int MODE>
__global__ __launch_bounds__(256) void k(float* G, const int* rows, int64_t n, float* O)
{
    const int lane = threadIdx.x & 63, sub = lane >> 5, kk = lane & 31;
    int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nw = (int64_t)gridDim.x * 4;
    unsigned xcc = 0;
    if (MODE >= 8) { asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc)); xcc &= 7; }
    for (int64_t b = w * 2 + sub; b < n; b += nw * 2) {
        int r = rows[b];
        if (MODE == 8 || MODE == 10) r = (r & ~7) | (int)xcc;           // row affine to this XCD
        if (MODE == 9) r = (r & ~7) | (int)((xcc + 1) & 7);           // row affine to ANOTHER XCD
        float* row = G + (size_t)r * 128;
        if (MODE == 0) {          // 4 x f32 atomics, 128 B contiguous per row per instr
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(row + kk + 32 * c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 1) {   // 2 x u64 atomics, 256 B contiguous per row per instr
            unsigned long long* r64 = (unsigned long long*)row;
            for (int c = 0; c < 2; ++c) __hip_atomic_fetch_add(r64 + kk + 32 * c, 0x100000001ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 2) {   // 4 x u32 atomics
            unsigned* r32 = (unsigned*)row;
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(r32 + kk + 32 * c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 3) {   // 2 x f64 atomics
            double* r64 = (double*)row;
            for (int c = 0; c < 2; ++c) __hip_atomic_fetch_add(r64 + kk + 32 * c, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 4) {   // plain RMW (non-atomic) for reference
            for (int c = 0; c < 4; ++c) row[kk + 32 * c] += 1.0f;
        } else if (MODE == 5) {   // f32 atomics, workgroup scope (XCD-local L2?)
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(row + kk + 32 * c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (MODE == 6) {   // plain stores only
            for (int c = 0; c < 4; ++c) row[kk + 32 * c] = 1.0f;
        } else if (MODE == 8 || MODE == 9) {
            for (int c = 0; c < 4; ++c) __hip_atomic_fetch_add(row + kk + 32 * c, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (MODE == 10) {
            for (int c = 0; c < 4; ++c) row[kk + 32 * c] += 1.0f;
        } else if (MODE == 11) {  // plain RMW with streaming (nt) loads and stores
            for (int c = 0; c < 4; ++c) { float v = __builtin_nontemporal_load(row + kk + 32 * c); __builtin_nontemporal_store(v + 1.0f, row + kk + 32 * c); }
        } else if (MODE == 12) {  // loads only (sum kept alive)
            float a = 0.f;
            for (int c = 0; c < 4; ++c) a += row[kk + 32 * c];
            if (a == 123.456f) row[0] = a;
        } else if (MODE == 13) {  // random row READ, the updated row WRITTEN in batch order (a log-structured table)
            float* orow = O + (size_t)b * 128;
            for (int c = 0; c < 4; ++c) { float v = __builtin_nontemporal_load(row + kk + 32 * c); __builtin_nontemporal_store(v + 1.0f, orow + kk + 32 * c); }
        } else if (MODE == 14) {  // rows read in batch order, written at random
            const float* irow = O + (size_t)b * 128;
            for (int c = 0; c < 4; ++c) { float v = __builtin_nontemporal_load(irow + kk + 32 * c); __builtin_nontemporal_store(v + 1.0f, row + kk + 32 * c); }
        } else if (MODE == 15) {  // both in batch order (a streaming copy with the same row layout)
            const float* irow = O + (size_t)b * 128; float* orow = G + (size_t)b * 128;
            for (int c = 0; c < 4; ++c) { float v = __builtin_nontemporal_load(irow + kk + 32 * c); __builtin_nontemporal_store(v + 1.0f, orow + kk + 32 * c); }
        } else if (MODE == 7) {   // packed bf16 atomics: 2 x (2 bf16 per dword)... use pk_add_f16 via builtin if available
            for (int c = 0; c < 2; ++c) {
                typedef _Float16 h2 __attribute__((ext_vector_type(2)));
                h2 v = {(_Float16)1.0f, (_Float16)1.0f};
                __builtin_amdgcn_global_atomic_fadd_v2f16((h2 __attribute__((address_space(1)))*)((h2*)row + kk + 32 * c), v);
            }
        }
    }
}

int main(int argc, char** argv)
{
    // atomic_bench [rows of the table] [row accesses per launch]; accesses <= rows: a permutation (every row once)
    const int64_t I = argc > 1 ? atoll(argv[1]) : 100000, n = argc > 2 ? atoll(argv[2]) : 2000000;
    printf("table %lld rows x 512 B, %lld row accesses per launch%s\n", (long long)I, (long long)n, n <= I ? " (each row at most once)" : "");
    float* G; int* rows; float* O;
    CK(hipMalloc(&O, (n < I ? n : I) * 128 * 4)); CK(hipMemset(O, 0, (n < I ? n : I) * 128 * 4));
    CK(hipMalloc(&G, I * 128 * 4)); CK(hipMemset(G, 0, I * 128 * 4));
    std::vector<int> h(n); uint64_t s = 88172645463325252ull;
    if (n <= I) {
        std::vector<int> perm(I);
        for (int64_t q = 0; q < I; ++q) perm[q] = (int)q;
        for (int64_t q = I - 1; q > 0; --q) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; std::swap(perm[q], perm[s % (q + 1)]); }
        for (int64_t q = 0; q < n; ++q) h[q] = perm[q];
    } else
    for (auto& x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (int)(s % I); }
    CK(hipMalloc(&rows, n * 4)); CK(hipMemcpy(rows, h.data(), n * 4, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const char* names[] = {"f32 atomic x4 (512B/row)", "u64 atomic x2 (512B/row)", "u32 atomic x4", "f64 atomic x2", "plain RMW x4", "f32 atomic wg-scope x4", "plain store x4", "pk f16 atomic x2 (256B/row)", "f32 atomic, row%8 == own XCD", "f32 atomic, row%8 == other XCD", "plain RMW, row%8 == own XCD", "plain RMW x4, nt", "loads only x4", "random read -> ordered write, nt", "ordered read -> random write, nt", "ordered read -> ordered write, nt"};
#define RUN(M) { for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<M>, dim3(2048), dim3(256), 0, 0, G, rows, n, O); \
    hipEventRecord(e0); for (int it = 0; it < 10; ++it) hipLaunchKernelGGL(k<M>, dim3(2048), dim3(256), 0, 0, G, rows, n, O); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10; printf("%-32s %8.1f us  %7.1f M rows/s\n", names[M], ms * 1e3, n / ms / 1e3); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    if (n <= I) { RUN(13) RUN(14) RUN(15) }
    return 0;
}
