// What does ONE column step of osd_kernel's 6-bit value solve cost when a wave runs it alone?  (the panel phase
// is one wave's dependent chain -- ballot, s_ff1, v_readlane, masked XOR -- and measured ~250 cycles per column inside the kernel,
// whatever the instruction count.)  The probe runs the same step in a loop on one wave per CU and prints s_memtime ticks per step
//   (a) the wave alone on its CU,
//   (b) with a second wave on the SAME SIMD spinning on vector instructions (what the kernel's partner wave does NOT do: it waits),
//   (c) with seven other waves hammering LDS (what the kernel's other waves do during the claims / absorb of their rows),
// and, for scale, (d) a chain of plain dependent v_xor and (e) of plain dependent s_xor.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/solve_chain_probe.hip -o /tmp/solve_chain_probe && /tmp/solve_chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ long long tick() { return (long long)__builtin_amdgcn_s_memtime(); }

// MODE 0: solve steps; 1: dependent v_xor chain; 2: dependent s_xor chain; 3: six INDEPENDENT v_xor per iteration; 4: six independent s_xor
template <int MODE>
__global__ __launch_bounds__(512) void probe(long long* out, unsigned int* sink, int nrep, int others) {
    __shared__ unsigned long long lds[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += blockDim.x) lds[i] = i * 0x9e3779b97f4a7c15ull;
    __syncthreads();
    if (wave == 0) {
        unsigned int X = (unsigned int)lane, Y = 0u, acc = 0u;
        unsigned long long avm = 0xfffffffffffffffeull;
        const long long t0 = tick();
        if (MODE == 0) {
            for (int r = 0; r < nrep; ++r) {
                X = (unsigned int)lane ^ (unsigned int)(r & 7);  // a fresh set of values every six steps
                Y = 0u;
#pragma unroll
                for (int j = 0; j < 6; ++j) {
                    unsigned int mh = (unsigned int)((int)(X << (31 - j)) >> 31);
                    asm volatile("" : "+v"(mh));
                    const unsigned long long cand = __ballot(mh != 0u) & avm;
                    if (cand) {
                        const int l = __ffsll((long long)cand) - 1;
                        const unsigned int ppk = (unsigned int)__builtin_amdgcn_readlane((int)X, l) & 0x3f3fu;
                        const unsigned int pn = ppk ^ (0x100u << j);
                        X = __builtin_amdgcn_bitop3_b32(pn, X, mh, 0x6c);
                        unsigned int my = (unsigned int)((int)(Y << (31 - j)) >> 31);
                        asm volatile("" : "+v"(my));
                        Y = __builtin_amdgcn_bitop3_b32(pn, Y, my, 0x6c);
                        unsigned int ml = lane == l ? ~0u : 0u;
                        asm volatile("" : "+v"(ml));
                        Y |= (ppk | ((unsigned int)j << 16) | (1u << 20)) & ml;
                    }
                }
                acc ^= X ^ Y;
            }
        } else if (MODE == 1) {
            unsigned int v = (unsigned int)lane;
            for (int r = 0; r < nrep; ++r) {
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(X));
            }
            acc = v;
        } else if (MODE == 3) {
            unsigned int v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3, v4 = lane + 4, v5 = lane + 5;
            for (int r = 0; r < nrep; ++r) {
                asm volatile("v_xor_b32 %0, %0, %6\n\tv_xor_b32 %1, %1, %6\n\tv_xor_b32 %2, %2, %6\n\tv_xor_b32 %3, %3, %6\n\tv_xor_b32 %4, %4, %6\n\tv_xor_b32 %5, %5, %6"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5) : "v"(X));
            }
            acc = v0 ^ v1 ^ v2 ^ v3 ^ v4 ^ v5;
        } else if (MODE == 4) {
            unsigned int s0 = (unsigned int)__builtin_amdgcn_readfirstlane(nrep), s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3, s4 = s0 + 4, s5 = s0 + 5;
            for (int r = 0; r < nrep; ++r) {
                asm volatile("s_xor_b32 %0, %0, 0x55\n\ts_xor_b32 %1, %1, 0x55\n\ts_xor_b32 %2, %2, 0x55\n\ts_xor_b32 %3, %3, 0x55\n\ts_xor_b32 %4, %4, 0x55\n\ts_xor_b32 %5, %5, 0x55"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5) : : "scc");
            }
            acc = s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5;
        } else {
            unsigned int sv = (unsigned int)__builtin_amdgcn_readfirstlane(nrep);
            for (int r = 0; r < nrep; ++r) {
#pragma unroll
                for (int j = 0; j < 6; ++j) asm volatile("s_xor_b32 %0, %0, 0x55" : "+s"(sv) : : "scc");
            }
            acc = sv;
        }
        const long long t1 = tick();
        if (lane == 0) out[blockIdx.x] = t1 - t0;
        sink[blockIdx.x * 64 + lane] = acc;
        lds[4095] = 1ull;  // stop flag for the others
    } else if (others == 1 && wave == 4) {
        // the partner wave on wave 0's SIMD (waves are dealt round-robin to the four SIMDs): vector work until the flag is up
        unsigned int v = (unsigned int)lane;
        volatile unsigned long long* f = &lds[4095];
        while (*f != 1ull) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v) : "v"(v));
        }
        sink[(gridDim.x + blockIdx.x) * 64 + lane] = v;
    } else if (others == 2) {
        // every other wave: LDS traffic until the flag is up
        unsigned long long a = 0ull;
        volatile unsigned long long* f = &lds[4095];
        int i = tid;
        while (*f != 1ull) {
#pragma unroll
            for (int j = 0; j < 16; ++j) { a ^= lds[(i + 64 * j) & 2047]; }
            i += 7;
        }
        sink[(2 * gridDim.x + blockIdx.x) * 64 + lane] = (unsigned int)a;
    }
}

int main() {
    const int ncu = 256, nrep = 20000;
    long long* d_out;
    unsigned int* d_sink;
    hipMalloc(&d_out, sizeof(long long) * ncu);
    hipMalloc(&d_sink, sizeof(unsigned int) * 64 * ncu * 4);
    std::vector<long long> h(ncu);
    auto run = [&](const char* what, int mode, int others) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, nrep, others);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, nrep, others);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, nrep, others);
            if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, nrep, others);
            if (mode == 4) hipLaunchKernelGGL(probe<4>, dim3(ncu), dim3(512), 0, 0, d_out, d_sink, nrep, others);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), d_out, sizeof(long long) * ncu, hipMemcpyDeviceToHost);
        double s = 0;
        for (long long v : h) s += (double)v;
        printf("%-78s %8.1f ticks per step (%d steps per CU)\n", what, s / ncu / (6.0 * nrep), 6 * nrep);
    };
    run("(a) solve step, the wave alone on its CU", 0, 0);
    run("(b) solve step, a second wave on the same SIMD issuing vector instructions", 0, 1);
    run("(c) solve step, the seven other waves reading LDS", 0, 2);
    run("(d) one dependent v_xor_b32 (x 6 per iteration), alone", 1, 0);
    run("(e) one dependent s_xor_b32 (x 6 per iteration), alone", 2, 0);
    run("(f) one of six INDEPENDENT v_xor_b32 per iteration, alone", 3, 0);
    run("(g) one of six independent s_xor_b32 per iteration, alone", 4, 0);
    return 0;
}
