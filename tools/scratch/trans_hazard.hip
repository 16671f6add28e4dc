// Probe behind DESIGN.md section 4.1 "a hazard found in round 4": how many wait states does gfx950 need between a transcendental and the VALU
// instruction that reads its result?  One wave per SIMD (the generated attention kernels' situation) and four; eight chained
// {v_exp_f32 x_i ; <separator> ; v_add_f32 acc, acc, x_i} per variant, the lane's sum compared with exp2 on the host.
//   hipcc --offload-arch=gfx950 -O2 tools/scratch/trans_hazard.hip -o /tmp/trans_hazard && /tmp/trans_hazard
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>
#define CHAIN(SEP)                                                                                                            \
    asm volatile("v_mov_b32 %0, 0\n"                                                                                           \
                 "v_exp_f32 %1, %1\n" SEP "v_add_f32 %0, %0, %1\n"                                                            \
                 "v_exp_f32 %2, %2\n" SEP "v_add_f32 %0, %0, %2\n"                                                            \
                 "v_exp_f32 %3, %3\n" SEP "v_add_f32 %0, %0, %3\n"                                                            \
                 "v_exp_f32 %4, %4\n" SEP "v_add_f32 %0, %0, %4\n"                                                            \
                 "v_exp_f32 %5, %5\n" SEP "v_add_f32 %0, %0, %5\n"                                                            \
                 "v_exp_f32 %6, %6\n" SEP "v_add_f32 %0, %0, %6\n"                                                            \
                 "v_exp_f32 %7, %7\n" SEP "v_add_f32 %0, %0, %7\n"                                                            \
                 "v_exp_f32 %8, %8\n" SEP "v_add_f32 %0, %0, %8\n"                                                            \
                 : "=&v"(acc), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7))
// the generated forward's form: the row sums alternate between two registers, so a sum does not wait for the one in front of it
#define CHAIN2(SEP)                                                                                                           \
    asm volatile("v_mov_b32 %0, 0\nv_mov_b32 %9, 0\n"                                                                        \
                 "v_exp_f32 %1, %1\n" SEP "v_add_f32 %0, %0, %1\n"                                                            \
                 "v_exp_f32 %2, %2\n" SEP "v_add_f32 %9, %9, %2\n"                                                            \
                 "v_exp_f32 %3, %3\n" SEP "v_add_f32 %0, %0, %3\n"                                                            \
                 "v_exp_f32 %4, %4\n" SEP "v_add_f32 %9, %9, %4\n"                                                            \
                 "v_exp_f32 %5, %5\n" SEP "v_add_f32 %0, %0, %5\n"                                                            \
                 "v_exp_f32 %6, %6\n" SEP "v_add_f32 %9, %9, %6\n"                                                            \
                 "v_exp_f32 %7, %7\n" SEP "v_add_f32 %0, %0, %7\n"                                                            \
                 "v_exp_f32 %8, %8\n" SEP "v_add_f32 %9, %9, %8\n"                                                            \
                 "s_nop 7\nv_add_f32 %0, %0, %9\n"                                                                            \
                 : "=&v"(acc), "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "=&v"(acc2))
template <int VAR>
__global__ void k(const float *in, float *out) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    float x0 = in[t * 8 + 0], x1 = in[t * 8 + 1], x2 = in[t * 8 + 2], x3 = in[t * 8 + 3], x4 = in[t * 8 + 4], x5 = in[t * 8 + 5], x6 = in[t * 8 + 6], x7 = in[t * 8 + 7];
    float acc, acc2;
    if constexpr (VAR == 5) CHAIN2("");
    else if constexpr (VAR == 6) CHAIN2("s_nop 0\n");
    else if constexpr (VAR == 7) CHAIN2("s_nop 1\n");
    else if constexpr (VAR == 0) CHAIN("");
    else if constexpr (VAR == 1) CHAIN("s_nop 0\n");
    else if constexpr (VAR == 2) CHAIN("s_nop 1\n");
    else if constexpr (VAR == 3) CHAIN("s_nop 3\n");
    else CHAIN("v_mov_b32 v255, v255\n");
    out[t] = acc;
}
template <int VAR>
void run(const char *name, const float *din, float *dout, const std::vector<float> &h, int threads, int blocks) {
    k<VAR><<<blocks, threads>>>(din, dout);
    std::vector<float> o((size_t)threads * blocks);
    hipMemcpy(o.data(), dout, o.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0, badlanes[64] = {0};
    for (size_t t = 0; t < o.size(); ++t) {
        double want = 0;
        for (int i = 0; i < 8; ++i) want += exp2((double)h[t * 8 + i]);
        if (!(fabs(o[t] - want) <= 1e-4 * (1 + fabs(want)))) { ++bad; ++badlanes[t % 64]; }
    }
    printf("%-28s %d threads/block: %ld of %zu lanes wrong", name, threads, bad, o.size());
    if (bad) { printf("  (lane ids wrong:"); for (int l = 0; l < 64; ++l) if (badlanes[l]) printf(" %d", l); printf(")"); }
    printf("\n");
}
int main() {
    const int threads_max = 256, blocks = 512;
    std::vector<float> h((size_t)threads_max * blocks * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = -4.f + 8.f * (float)((i * 2654435761u) % 1000) / 1000.f;
    float *din, *dout;
    hipMalloc(&din, h.size() * 4); hipMalloc(&dout, (size_t)threads_max * blocks * 4);
    hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int threads : {64, 256}) {
        run<0>("adjacent", din, dout, h, threads, blocks);
        run<1>("s_nop 0 (1 wait state)", din, dout, h, threads, blocks);
        run<2>("s_nop 1 (2 wait states)", din, dout, h, threads, blocks);
        run<3>("s_nop 3", din, dout, h, threads, blocks);
        run<4>("an unrelated VALU between", din, dout, h, threads, blocks);
        run<5>("two sums: adjacent", din, dout, h, threads, blocks);
        run<6>("two sums: s_nop 0", din, dout, h, threads, blocks);
        run<7>("two sums: s_nop 1", din, dout, h, threads, blocks);
    }
    return 0;
}
