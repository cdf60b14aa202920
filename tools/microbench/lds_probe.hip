// LDS read-rate probe for gfx950: cycles per wave-level ds_read_b64 / ds_read_b128 for several address patterns.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/lds_probe.hip -o build/lds_probe && build/lds_probe
// One 1024-thread workgroup per CU (16 waves, like osd_large_kernel); every wave issues NREP dependent-free reads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef const volatile __attribute__((address_space(3))) unsigned long long* lds64;

template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long* out, long long* ticks, const unsigned* idxsrc, int nrep) {
    extern __shared__ unsigned long long lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 8192; i += 1024) lds[i] = i * 0x9E3779B97F4A7C15ull;
    __syncthreads();
    unsigned r = idxsrc[blockIdx.x * 1024 + tid];  // random per lane
    unsigned off;
    // byte offsets
    if (MODE == 0) off = (tid & 63) * 8;                         // linear, conflict-free, all 64 banks
    else if (MODE == 1) off = (r & 15) * 8;                      // random entry of a 128-byte table (16 x 8 B)
    else if (MODE == 2) off = (r & 15) * 16;                     // random entry, entries spread over 256 bytes
    else if (MODE == 3) off = (r & 31) * 8;                      // random entry of a 256-byte table (32 x 8 B)
    else if (MODE == 4) off = (r & 255) * 8;                     // random entry of a 2 KB table (256 x 8 B)
    else if (MODE == 5) off = 0;                                 // full broadcast
    else if (MODE == 6) off = (tid & 63) * 16;                   // linear b128
    else off = (r & 15) * 8 + ((tid >> 5) & 1) * 128;            // 16 entries, the two half-waves use different 128 B tables
    lds64 p = (lds64)((const char*)lds + off);
    unsigned long long a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < nrep; ++i) {
        if (MODE == 6) {
            typedef const volatile __attribute__((address_space(3))) ulonglong2* lds128;
            lds128 q = (lds128)((const char*)lds + off);
            ulonglong2 v0 = {q[0].x, q[0].y}, v1 = {q[64].x, q[64].y};
            a0 ^= v0.x; a1 ^= v0.y; a2 ^= v1.x; a3 ^= v1.y;
        } else {
            a0 ^= p[0 * 256]; a1 ^= p[1 * 256]; a2 ^= p[2 * 256]; a3 ^= p[3 * 256];
            a0 ^= p[4 * 256]; a1 ^= p[5 * 256]; a2 ^= p[6 * 256]; a3 ^= p[7 * 256];
        }
    }
    __syncthreads();
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + tid] = a0 ^ a1 ^ a2 ^ a3;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

// write probes: MODE 0 = b64 linear, 1 = b128 linear, 2 = b32 linear, 3 = b64 at stride 16 B (2-way), 4 = 2 x b32 split of a b64
template <int MODE>
__global__ __launch_bounds__(1024) void wprobe(unsigned long long* out, long long* ticks, int nrep) {
    extern __shared__ unsigned long long lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    typedef volatile __attribute__((address_space(3))) unsigned long long* w64;
    typedef volatile __attribute__((address_space(3))) unsigned int* w32;
    typedef volatile __attribute__((address_space(3))) ulonglong2* w128;
    char* base = (char*)lds + (tid >> 6) * 4096;  // 4 KB per wave
    unsigned long long v = tid;
    __syncthreads();
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
    for (int i = 0; i < nrep; ++i) {
        v += i;
        if (MODE == 0) {
            w64 p = (w64)(base + lane * 8);
            p[0] = v; p[64] = v; p[128] = v; p[192] = v; p[256] = v; p[320] = v; p[384] = v; p[448] = v;
        } else if (MODE == 1) {
            w128 p = (w128)(base + lane * 16);
            ulonglong2 x; x.x = v; x.y = v;
            p[0].x = x.x; p[0].y = x.y; p[64].x = x.x; p[64].y = x.y; p[128].x = x.x; p[128].y = x.y; p[192].x = x.x; p[192].y = x.y;
        } else if (MODE == 2) {
            w32 p = (w32)(base + lane * 4);
            p[0] = (unsigned)v; p[64] = (unsigned)v; p[128] = (unsigned)v; p[192] = (unsigned)v;
            p[256] = (unsigned)v; p[320] = (unsigned)v; p[384] = (unsigned)v; p[448] = (unsigned)v;
        } else if (MODE == 3) {
            w64 p = (w64)(base + (lane & 31) * 16 + (lane >> 5) * 8);
            p[0] = v; p[64] = v; p[128] = v; p[192] = v; p[256] = v; p[320] = v; p[384] = v; p[448] = v;
        }
    }
    __syncthreads();
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + tid] = lds[tid] + v;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main() {
    int ncu = 256;
    const int nrep = 4096;
    unsigned long long* out; long long* ticks; unsigned* idx;
    hipMalloc(&out, ncu * 1024 * 8); hipMalloc(&ticks, ncu * 8); hipMalloc(&idx, ncu * 1024 * 4);
    std::vector<unsigned> h(ncu * 1024);
    srand(1);
    for (auto& v : h) v = rand();
    hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<long long> t(ncu);
    const char* names[8] = {"b64 linear (64 banks)", "b64 random in 128 B table", "b64 random, 16 entries over 256 B",
                            "b64 random in 256 B table (32 entries)", "b64 random in 2 KB table (256 entries)",
                            "b64 broadcast", "b128 linear", "b64 random 16-entry, half-waves on different tables"};
#define RUN(M)                                                                                         \
    {                                                                                                  \
        hipLaunchKernelGGL(probe<M>, dim3(ncu), dim3(1024), 65536, 0, out, ticks, idx, nrep);          \
        hipDeviceSynchronize();                                                                        \
        hipLaunchKernelGGL(probe<M>, dim3(ncu), dim3(1024), 65536, 0, out, ticks, idx, nrep);          \
        hipDeviceSynchronize();                                                                        \
        hipMemcpy(t.data(), ticks, ncu * 8, hipMemcpyDeviceToHost);                                    \
        double s = 0; for (auto v : t) s += v;                                                         \
        const double per = s / ncu / ((double)nrep * (M == 6 ? 2 : 8) * 16);                           \
        printf("%-52s %.2f ticks per wave-level read (CU-wide, 16 waves)\n", names[M], per);          \
    }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    const char* wnames[4] = {"write b64 linear", "write b128 linear (asm may split)", "write b32 linear", "write b64 interleaved halves"};
#define WRUN(M, NW, BYTES)                                                                             \
    {                                                                                                  \
        hipLaunchKernelGGL(wprobe<M>, dim3(ncu), dim3(1024), 65536, 0, out, ticks, nrep);              \
        hipDeviceSynchronize();                                                                        \
        hipLaunchKernelGGL(wprobe<M>, dim3(ncu), dim3(1024), 65536, 0, out, ticks, nrep);              \
        hipDeviceSynchronize();                                                                        \
        hipMemcpy(t.data(), ticks, ncu * 8, hipMemcpyDeviceToHost);                                    \
        double s = 0; for (auto v : t) s += v;                                                         \
        const double per = s / ncu / ((double)nrep * NW * 16);                                         \
        printf("%-52s %.2f ticks per wave-level write of %d B (CU-wide, 16 waves)\n", wnames[M], per, BYTES); \
    }
    WRUN(0, 8, 512) WRUN(1, 4, 1024) WRUN(2, 8, 256) WRUN(3, 8, 512)
    return 0;
}
