// Stand-alone check of the two hardware layouts the split-product minibatch kernel (csrc/ppo_trunk_bx.hip) relies on:
//  (1) v_mfma_f32_32x32x16_bf16: lane (li = lane & 31, lh = lane >> 5) supplies A[i = li][k = 8 lh + e], B[k = 8 lh + e][j = li], e = 0..7
//  (2) ds_read_b64_tr_b16 over a row-major [rows][LD] 16-bit plane: lane q of a 16-lane group passes the address of
//      (row r0 + q / 4, col c0 + 4 (q % 4)); lane i of the group receives column c0 + i of rows r0 .. r0 + 3
// build: hipcc --offload-arch=gfx950 -O2 tools/csrc/probe_bf16x3.hip -o gpurun_out/probe_bf16x3 ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__global__ void mfma_probe(const unsigned short* A, const unsigned short* B, float* C) {   // A [32][16], B [16][32] as bf16 bits
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = __builtin_bit_cast(__bf16, A[li * 16 + 8 * lh + e]);
        b[e] = __builtin_bit_cast(__bf16, B[(8 * lh + e) * 32 + li]);
    }
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * lh) * 32 + li] = acc[r];
}

constexpr int LD = 136;
__global__ void tr_probe(unsigned short* out, int r0, int c0) {         // out[lane][8]: rows r0 + 8 lh + 0..7 of column c0 + li
    __shared__ __attribute__((aligned(16))) unsigned short pl[64 * LD];
    for (int e = threadIdx.x; e < 64 * LD; e += 64) pl[e] = (unsigned short)((e / LD) * 256 + (e % LD));
    __syncthreads();
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5, i = li & 15;
    for (int rd = 0; rd < 2; ++rd) {
        const unsigned short* at = pl + (r0 + 8 * lh + 4 * rd + (i >> 2)) * LD + c0 + 16 * (li >> 4) + 4 * (i & 3);
        s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)at);
        for (int j = 0; j < 4; ++j) out[lane * 8 + 4 * rd + j] = (unsigned short)t[j];
    }
}

// (3) how v_mfma_f32_32x32x16_bf16 rounds acc + sum of products: one product p = a * b (exact in bf16 x bf16) onto C = c
__global__ void round_probe(float c, float a, float b, int nprod, float* out) {
    const int lane = threadIdx.x;
    bf16x8 av, bv;
    for (int e = 0; e < 8; ++e) { av[e] = (__bf16)0.f; bv[e] = (__bf16)0.f; }
    if (lane < 32) for (int e = 0; e < nprod && e < 8; ++e) { av[e] = (__bf16)a; bv[e] = (__bf16)b; }   // k = 0..nprod-1 (lh = 0)
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}
__global__ void round_probe_f32(float c, float a, float b, float* out) {
    const int lane = threadIdx.x;
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = c;
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(lane < 32 ? a : 0.f, lane < 32 ? b : 0.f, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
    std::vector<unsigned short> A(32 * 16), B(16 * 32);
    srand(7);
    for (auto& v : A) v = f2bf((float)(rand() % 200 - 100) / 16.f);
    for (auto& v : B) v = f2bf((float)(rand() % 200 - 100) / 32.f);
    unsigned short *dA, *dB, *dO; float* dC;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&dO, 64 * 8 * 2);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC);
    std::vector<float> C(32 * 32);
    hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += (double)bf2f(A[i * 16 + k]) * bf2f(B[k * 32 + j]);
        if (fabs(s - C[i * 32 + j]) > 1e-3) ++bad;
    }
    printf("mfma_32x32x16_bf16 layout: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    int rc = bad != 0;
    const int cases[3][2] = {{0, 0}, {16, 32}, {48, 96}};
    for (auto& cs : cases) {
        hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, dO, cs[0], cs[1]);
        std::vector<unsigned short> O(64 * 8);
        hipMemcpy(O.data(), dO, O.size() * 2, hipMemcpyDeviceToHost);
        int b2 = 0;
        for (int lane = 0; lane < 64; ++lane) for (int e = 0; e < 8; ++e) {
            const int row = cs[0] + 8 * (lane >> 5) + e, col = cs[1] + (lane & 31);
            if (O[lane * 8 + e] != (unsigned short)(row * 256 + col)) { if (b2 < 4) printf("  lane %d e %d: got row %d col %d want row %d col %d\n", lane, e, O[lane * 8 + e] >> 8, O[lane * 8 + e] & 255, row, col); ++b2; }
        }
        printf("ds_read_b64_tr_b16 (r0 %d, c0 %d): %s (%d mismatches)\n", cs[0], cs[1], b2 ? "WRONG" : "ok", b2);
        rc |= b2 != 0;
    }
    {
        float* dR; hipMalloc(&dR, 4);
        struct { float c, a, b; int n; const char* what; } rc[] = {
            {1.f, 0x1p-13f, 0x1p-13f, 1, "+1 + 2^-26        (RNE +1 | to-zero +1 | to -inf +1)"},
            {1.f, -0x1p-13f, 0x1p-13f, 1, "+1 - 2^-26        (RNE +1 | to-zero 1-2^-24 | to -inf 1-2^-24)"},
            {-1.f, 0x1p-13f, 0x1p-13f, 1, "-1 + 2^-26        (RNE -1 | to-zero -(1-2^-24) | to -inf -1)"},
            {1.f, 1.5f, 0x1p-24f, 1, "+1 + 0.75 ulp     (RNE 1+2^-23 | to-zero +1 | to -inf +1)"},
            {-1.f, -1.5f, 0x1p-24f, 1, "-1 - 0.75 ulp     (RNE -(1+2^-23) | to-zero -1 | to -inf -(1+2^-23))"},
            {1.f, 0x1p-13f, 0x1p-14f, 8, "+1 + 8 x 2^-27 = +1 + 2^-24 = half an ulp, in 8 products (exact sum first: tie -> +1; product by product to-zero: +1)"},
            {1.f, 1.5f, 0x1p-27f, 8, "+1 + 8 x 1.5 x 2^-27 = +1 + 0.75 ulp in 8 products (sum first: RNE 1+2^-23)"},
        };
        for (auto& t : rc) {
            float r = 0, r32 = 0;
            hipLaunchKernelGGL(round_probe, dim3(1), dim3(64), 0, 0, t.c, t.a, t.b, t.n, dR);
            hipMemcpy(&r, dR, 4, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(round_probe_f32, dim3(1), dim3(64), 0, 0, t.c, t.a, t.b * (float)t.n, dR);
            hipMemcpy(&r32, dR, 4, hipMemcpyDeviceToHost);
            printf("round %s: bf16 mfma %.9g (%a)   f32 mfma %.9g (%a)\n", t.what, r, r, r32, r32);
        }
    }
    if (hipDeviceSynchronize() != hipSuccess) { printf("device error\n"); return 2; }
    return rc;
}
