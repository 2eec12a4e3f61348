// Minibatch index generation on the device: `n_perm` independent pseudo-random permutations of [0, N), one per epoch
// (np.random.shuffle(indexes) per epoch in the reference, core/on_policy.py:194-204; the RNG stream itself cannot be
// matched -- SURVEY.md section 7 -- parity tests pin the indices instead).  Each permutation is a keyed bijection of
// [0, 2^b) (b = ceil(log2 N), alternating unbalanced Feistel network, 8 rounds of a 32-bit mixing function) restricted
// to [0, N) by cycle walking, so every output is one independent thread: one launch, no sort, capturable in a hipGraph
// (the key comes from a device counter that the host bumps once per update phase with xrl_counter_add).
#include "common.h"
#include "rng.h"

namespace xrl {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {       // murmur3 finaliser
    x ^= x >> 16; x *= 0x85EBCA6Bu; x ^= x >> 13; x *= 0xC2B2AE35u; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) random_permutation_kernel(int64_t* __restrict__ out, int n_perm, int64_t N,
                                                                 int64_t take, uint64_t seed, uint32_t counter,
                                                                 const uint32_t* __restrict__ counter_dev) {
    const uint32_t ctr = counter + (counter_dev ? *counter_dev : 0u);
    int b = 1;
    while (((int64_t)1 << b) < N) ++b;
    const int bl = b / 2, br = b - bl;
    const uint32_t ml = (1u << bl) - 1u, mr = (1u << br) - 1u;
    const int64_t total = (int64_t)n_perm * take;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int e = (int)(i / take);
        const int64_t j = i - (int64_t)e * take;
        uint32_t kk[4];
        philox4x32(seed, ctr, (uint32_t)e, 0x5045524Du, kk);            // 128-bit key of (seed, update phase, epoch)
        uint32_t x = (uint32_t)j;
        do {
            uint32_t L = x >> br, R = x & mr;
#pragma unroll
            for (int r = 0; r < 8; r += 2) {
                L = (L ^ mix32(R + kk[r & 3] + 0x9E3779B9u * (uint32_t)(r + 1))) & ml;
                R = (R ^ mix32(L + kk[(r + 1) & 3] + 0x9E3779B9u * (uint32_t)(r + 2))) & mr;
            }
            x = (L << br) | R;
        } while ((int64_t)x >= N);
        out[i] = (int64_t)x;
    }
}

// Uniform replay sampling with replacement (DummyOffPolicyBuffer.sample, memory_tools.py:376-377; MARL_OffPolicyBuffer
// .sample, memory_tools_marl.py:753-754): env = choice(n_envs), step = choice(size), flat index env * n_size + step.
// `size` is read from device memory so that a captured graph follows the filling ring.
__global__ void __launch_bounds__(256) sample_replay_kernel(int64_t* __restrict__ out, int bs, int n_envs, int n_size,
                                                            const int32_t* __restrict__ size_dev, uint64_t seed, uint32_t counter,
                                                            const uint32_t* __restrict__ counter_dev) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= bs) return;
    const uint32_t ctr = counter + (counter_dev ? *counter_dev : 0u);
    int size = *size_dev;
    size = size < 1 ? 1 : (size > n_size ? n_size : size);
    uint32_t r[4];
    philox4x32(seed, (uint32_t)b, ctr, 0x53414D50u, r);
    const int env = (int)(((uint64_t)r[0] * (uint64_t)n_envs) >> 32);        // floor(u * n), u = r / 2^32
    const int step = (int)(((uint64_t)r[1] * (uint64_t)size) >> 32);
    out[b] = (int64_t)env * n_size + step;
}

}  // namespace xrl

extern "C" int xrl_sample_replay_indices(int64_t* out, int bs, int n_envs, int n_size, const int32_t* size_dev, uint64_t seed,
                                         uint32_t counter, const uint32_t* counter_dev, xrl_stream_t stream) {
    XRL_CHECK_ARG(out && size_dev && bs > 0 && n_envs > 0 && n_size > 0);
    hipLaunchKernelGGL(xrl::sample_replay_kernel, dim3((bs + 255) / 256), dim3(256), 0, xrl::as_stream(stream), out, bs, n_envs,
                       n_size, size_dev, seed, counter, counter_dev);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_random_permutation(int64_t* out, int n_perm, int64_t N, int64_t take, uint64_t seed, uint32_t counter,
                                      const uint32_t* counter_dev, xrl_stream_t stream) {
    XRL_CHECK_ARG(out && n_perm > 0 && N > 1 && N <= ((int64_t)1 << 31) && take > 0 && take <= N);
    const int64_t total = (int64_t)n_perm * take;
    int nb = (int)((total + 255) / 256);
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(xrl::random_permutation_kernel, dim3(nb), dim3(256), 0, xrl::as_stream(stream), out, n_perm, N, take,
                       seed, counter, counter_dev);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
