// Prioritized replay: PerOffPolicyBuffer's per-env sum / min segment trees (xuance/common/memory_tools.py:471-598,
// segtree_tool.py:24-230) on the device.  One pair of trees per env, [n_envs][2 * capacity] float64 (the reference keeps
// Python floats; under the NumPy < 2 it pins every priority is promoted to float64), node 1 = root, leaves at
// capacity + i.  All three operations are tiny and latency-bound (n_envs independent trees, log2(capacity) levels):
// one thread per env (store, update -- sequential semantics per env: later duplicates win) or per sample (sampling).
#include "common.h"

namespace xrl {

__device__ __forceinline__ void tree_set(double* sum, double* mn, int cap, int idx, double val) {   // __setitem__ :98-113
    int i = idx + cap;
    sum[i] = val; mn[i] = val;
    for (i >>= 1; i >= 1; i >>= 1) {
        sum[i] = sum[2 * i] + sum[2 * i + 1];
        mn[i] = fmin(mn[2 * i], mn[2 * i + 1]);
    }
}

// reduce(0, end_excl) of the sum tree with the association order of _reduce_helper (:41-63): for a prefix range the
// recursion returns v[left child] + (recursion into the right child), nested to the right.
__device__ __forceinline__ double tree_prefix_sum(const double* sum, int cap, int end_incl) {
    double lefts[32];
    int nl = 0, node = 1, ns = 0, ne = cap - 1;
    while (!(end_incl == ne)) {
        const int mid = (ns + ne) >> 1;
        if (end_incl <= mid) { node = 2 * node; ne = mid; }
        else { lefts[nl++] = sum[2 * node]; node = 2 * node + 1; ns = mid + 1; }
    }
    double acc = sum[node];
    for (int i = nl - 1; i >= 0; --i) acc = lefts[i] + acc;
    return acc;
}

__global__ void per_store_kernel(double* sum, double* mn, const double* max_priority, int ptr, double alpha, int n_envs,
                                 int cap) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    const double v = pow(max_priority[e], alpha);                                    // memory_tools.py:536-538
    tree_set(sum + (size_t)e * 2 * cap, mn + (size_t)e * 2 * cap, cap, ptr, v);
}

// grid: n_envs blocks, k threads (k = batch_size / n_envs samples per env, memory_tools.py:542-565)
__global__ void per_sample_kernel(const double* sum_all, const double* mn_all, const double* uniforms, int size, double beta,
                                  int n_envs, int n_size, int cap, int k, int64_t* step_choices, double* weights,
                                  int64_t* flat_idx) {
    const int e = blockIdx.x, i = threadIdx.x;
    if (i >= k) return;
    const double* sum = sum_all + (size_t)e * 2 * cap;
    const double* mn = mn_all + (size_t)e * 2 * cap;
    const double p_total = tree_prefix_sum(sum, cap, size - 2);                      // sum(0, size - 1): end exclusive (:502)
    const double every = p_total / (double)k;                                        // :503
    double mass = uniforms[(size_t)e * k + i] * every + (double)i * every;           // :505
    int idx = 1;                                                                     // find_prefixsum_idx (segtree_tool.py:160-170)
    while (idx < cap) {
        if (sum[2 * idx] > mass) idx = 2 * idx;
        else { mass -= sum[2 * idx]; idx = 2 * idx + 1; }
    }
    idx -= cap;
    const double total = sum[1];
    const double p_min = mn[1] / total;                                              // :550
    const double scale = pow((double)size, -beta);
    const double max_weight = p_min * scale;                                         // :551
    const double w = (sum[cap + idx] / total) * scale / max_weight;                  // :553-555
    step_choices[(size_t)e * k + i] = idx;
    weights[(size_t)e * k + i] = w;
    if (flat_idx) flat_idx[(size_t)e * k + i] = (int64_t)e * n_size + idx;           // env-major flat index for xrl_soa_gather
}

__global__ void per_update_kernel(double* sum, double* mn, double* max_priority, const int64_t* idxes, const float* prio,
                                  double alpha, int n_envs, int cap, int k) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    double* s = sum + (size_t)e * 2 * cap;
    double* m = mn + (size_t)e * 2 * cap;
    double mp = max_priority[e];
    for (int i = 0; i < k; ++i) {                                                    // :588-597, in order
        double p = (double)prio[(size_t)e * k + i];
        if (p == 0.0) p += 1e-8;
        tree_set(s, m, cap, (int)idxes[(size_t)e * k + i], pow(p, alpha));
        mp = fmax(mp, p);
    }
    max_priority[e] = mp;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_per_store(double* sum_tree, double* min_tree, const double* max_priority, int ptr, double alpha,
                             int n_envs, int capacity, xrl_stream_t stream) {
    XRL_CHECK_ARG(sum_tree && min_tree && max_priority && n_envs > 0 && capacity > 0 && (capacity & (capacity - 1)) == 0);
    XRL_CHECK_ARG(ptr >= 0 && ptr < capacity);
    hipLaunchKernelGGL(per_store_kernel, dim3((n_envs + 63) / 64), dim3(64), 0, as_stream(stream), sum_tree, min_tree,
                       max_priority, ptr, alpha, n_envs, capacity);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_per_sample(const double* sum_tree, const double* min_tree, const double* uniforms, int size, double beta,
                              int n_envs, int n_size, int capacity, int per_env, int64_t* step_choices, double* weights,
                              int64_t* flat_idx, xrl_stream_t stream) {
    XRL_CHECK_ARG(sum_tree && min_tree && uniforms && step_choices && weights && n_envs > 0 && per_env > 0 && per_env <= 1024);
    XRL_CHECK_ARG(capacity > 0 && (capacity & (capacity - 1)) == 0 && size >= 2 && size <= capacity && beta > 0);
    hipLaunchKernelGGL(per_sample_kernel, dim3(n_envs), dim3((per_env + 63) / 64 * 64), 0, as_stream(stream), sum_tree, min_tree,
                       uniforms, size, beta, n_envs, n_size, capacity, per_env, step_choices, weights, flat_idx);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_per_update_priorities(double* sum_tree, double* min_tree, double* max_priority, const int64_t* idxes,
                                         const float* priorities, double alpha, int n_envs, int capacity, int per_env,
                                         xrl_stream_t stream) {
    XRL_CHECK_ARG(sum_tree && min_tree && max_priority && idxes && priorities && n_envs > 0 && per_env > 0);
    XRL_CHECK_ARG(capacity > 0 && (capacity & (capacity - 1)) == 0);
    hipLaunchKernelGGL(per_update_kernel, dim3((n_envs + 63) / 64), dim3(64), 0, as_stream(stream), sum_tree, min_tree,
                       max_priority, idxes, priorities, alpha, n_envs, capacity, per_env);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
