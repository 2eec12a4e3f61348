// Episode replay buffer of the recurrent multi-agent path: MARL_OffPolicyBuffer_RNN
// (xuance/common/memory_tools_marl.py:770-996) on the device.  Per-env staging rows ("episode_data") collect a running
// episode; a finished episode is copied as a whole row into the ring ("data"); sampling gathers whole episodes.
// Layout: every field is [episode][slot][row] with the agents of a step contiguous inside `row` (parameter sharing: the
// agents are homogeneous), so a step write is one contiguous row per field and an episode is one contiguous block; the
// gather writes the learner's TIME-MAJOR batch ([slot][episode][row]) directly.  All HBM-bound, 4- or 16-byte units.
#include "common.h"
#include "rng.h"

namespace xrl {

constexpr int EP_MAX_FIELDS = 12;
struct EpPack {
    void* a[EP_MAX_FIELDS];
    const void* b[EP_MAX_FIELDS];
    const void* c[EP_MAX_FIELDS];
    const void* d[EP_MAX_FIELDS];
    int row_bytes[EP_MAX_FIELDS];
    int slots[EP_MAX_FIELDS];
    int flags[EP_MAX_FIELDS];
    int n;
};

// staging[env][steps[env]] <- src[env]   (store, :904-921).  grid (n_envs, n_fields)
__global__ void __launch_bounds__(64) episode_store_kernel(EpPack f, const int32_t* __restrict__ steps) {
    const int env = blockIdx.x, fi = blockIdx.y;
    const int rw = f.row_bytes[fi] >> 2;
    const int st = steps[env];
    if (st < 0 || st >= f.slots[fi]) return;
    uint32_t* d = reinterpret_cast<uint32_t*>(f.a[fi]) + ((size_t)env * f.slots[fi] + st) * rw;
    const uint32_t* s = reinterpret_cast<const uint32_t*>(f.b[fi]) + (size_t)env * rw;
    for (int i = threadIdx.x; i < rw; i += 64) d[i] = s[i];
}

// finish_path + store_episodes (:923-968) for every finished env, in env order.  grid (n_envs, n_fields), 256 threads
// steps != NULL: the step's store (episode_store_kernel's statements, field d -> staging) happens first, in the same block that
// then closes the episode -- one launch per vector step for both (xrl_episode_store_finish).
__device__ __forceinline__ void episode_finish_body(const EpPack& f, const float* __restrict__ gate, const float* __restrict__ done,
                                                    const int32_t* __restrict__ end_step, const int32_t* __restrict__ ptr_size,
                                                    int buffer_size, const int32_t* __restrict__ steps) {
    const int env = blockIdx.x, fi = blockIdx.y;
    if (steps && f.d[fi]) {
        const int rw0 = f.row_bytes[fi] >> 2, st = steps[env];
        if (st >= 0 && st < f.slots[fi]) {
            uint32_t* d = reinterpret_cast<uint32_t*>(const_cast<void*>(f.b[fi])) + ((size_t)env * f.slots[fi] + st) * rw0;
            const uint32_t* s = reinterpret_cast<const uint32_t*>(f.d[fi]) + (size_t)env * rw0;
            for (int i = threadIdx.x; i < rw0; i += 256) d[i] = s[i];
        }
        __syncthreads();                                                 // (this block is the staging row's only writer and reader)
    }
    if ((gate && *gate == 0.f) || done[env] == 0.f) return;
    int rank = 0;
    for (int j = 0; j < env; ++j) rank += done[j] != 0.f;                // this env's position among the finished ones
    const int dst_ep = (ptr_size[0] + rank) % buffer_size;               // self.ptr advances once per stored episode
    const int rw = f.row_bytes[fi] >> 2, slots = f.slots[fi];
    uint32_t* stage = reinterpret_cast<uint32_t*>(const_cast<void*>(f.b[fi])) + (size_t)env * slots * rw;
    uint32_t* ring = reinterpret_cast<uint32_t*>(f.a[fi]) + (size_t)dst_ep * slots * rw;
    const int es = end_step[env];
    const uint32_t* term = f.c[fi] ? reinterpret_cast<const uint32_t*>(f.c[fi]) + (size_t)env * rw : nullptr;
    const int n = slots * rw;
    // (eight words requested before the first store, as in the gather below: a finished episode is up to 5 490 words per block)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            v[u] = 0u;
            if (i < n) v[u] = (term && i / rw == es) ? term[i - es * rw] : stage[i];   // terminal obs / state / avail_actions (:931-937)
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            if (i < n) {
                if (term && i / rw == es) stage[i] = v[u];               // the reference writes it into episode_data first
                ring[i] = v[u];
                if (f.flags[fi] & 1) stage[i] = 0u;                      // `filled` of this env is cleared (:921)
            }
        }
    }
}

__global__ void __launch_bounds__(256) episode_finish_kernel(EpPack f, const float* __restrict__ gate, const float* __restrict__ done,
                                                             const int32_t* __restrict__ end_step,
                                                             const int32_t* __restrict__ ptr_size, int n_envs,
                                                             int buffer_size, const int32_t* __restrict__ steps) {
    episode_finish_body(f, gate, done, end_step, ptr_size, buffer_size, steps);
}

__global__ void episode_advance_kernel(const float* __restrict__ gate, const float* __restrict__ done, int32_t* __restrict__ ptr_size, int n_envs,
                                       int buffer_size) {
    if (gate && *gate == 0.f) return;
    int c = 0;
    for (int j = 0; j < n_envs; ++j) c += done[j] != 0.f;
    ptr_size[0] = (ptr_size[0] + c) % buffer_size;
    const int s = ptr_size[1] + c;
    ptr_size[1] = s < buffer_size ? s : buffer_size;
}

// dst[slot][b][row] = ring[idx[b]][slot][row]   (sample, :970-996, transposed to time-major).  grid (B, n_fields)
__global__ void __launch_bounds__(256) episode_gather_kernel(EpPack f, const int64_t* __restrict__ idx, int B) {
    const int b = blockIdx.x, fi = blockIdx.y;
    const int rw = f.row_bytes[fi] >> 2, slots = f.slots[fi];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(f.b[fi]) + (size_t)idx[b] * slots * rw;
    uint32_t* d = reinterpret_cast<uint32_t*>(f.a[fi]);
    const int n = slots * rw;
    for (int i = threadIdx.x; i < n; i += 256) {
        const int t = i / rw, w = i - t * rw;
        d[((size_t)t * B + b) * rw + w] = s[i];
    }
}

// The same with the uniform draw inside (xrl_sample_replay_indices' stream with n_envs = 1: episode = floor(u * size), u from
// Philox(seed, b, counter + *counter_dev)): every workgroup of episode b forms the same index itself, field 0's writes it to idx_out.
// One launch less per update of the recurrent learners (a launch costs ~5 us at these sizes whatever it does).
__global__ void __launch_bounds__(256) episode_gather_sampled_kernel(EpPack f, int64_t* __restrict__ idx_out, int B, int n_size,
                                                                    const int32_t* __restrict__ size_dev, uint64_t seed, uint32_t counter,
                                                                    const uint32_t* __restrict__ counter_dev) {
    const int b = blockIdx.x, fi = blockIdx.y;
    const uint32_t ctr = counter + (counter_dev ? *counter_dev : 0u);
    int size = *size_dev;
    size = size < 1 ? 1 : (size > n_size ? n_size : size);
    uint32_t r[4];
    philox4x32(seed, (uint32_t)b, ctr, 0x53414D50u, r);
    const int64_t ep = (int64_t)(((uint64_t)r[1] * (uint64_t)size) >> 32);   // (n_envs = 1: env = 0, flat index = step)
    if (fi == 0 && threadIdx.x == 0) idx_out[b] = ep;
    const int rw = f.row_bytes[fi] >> 2, slots = f.slots[fi];
    const uint32_t* s = reinterpret_cast<const uint32_t*>(f.b[fi]) + (size_t)ep * slots * rw;
    uint32_t* d = reinterpret_cast<uint32_t*>(f.a[fi]);
    const int n = slots * rw;
    // (eight loads requested before the first store: an episode of the obs field is 5 490 words = 21 trips of this loop per thread, and
    //  as a plain loop every trip was a global round trip of its own)
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * 256) {
        uint32_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int i = i0 + u * 256; v[u] = i < n ? s[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 256;
            if (i < n) { const int t = i / rw, w = i - t * rw; d[((size_t)t * B + b) * rw + w] = v[u]; }
        }
    }
}

// Loop control of OffPolicyMARLAgents.run_episodes (off_policy_marl.py:464-546) kept on the device, so that the host may
// enqueue vector step s + 1 before it has read the outcome of step s.  Launched at the END of every captured vector step: it
// settles the step exactly as the host loop does after env.step -- episodes finished so far, `current_step += episode_step`
// of the episodes that ended (:532), `_update_explore_factor()` (:197-204, incl. its one-call undershoot below end_greedy)
// -- and decides whether the NEXT step still belongs to the call (`episodes < n_episodes`, :464).  A step launched after the
// call is over is dry: the captured step multiplies `done` by active_f (no episode is closed into the ring) and adds
// active_i to its RNG step counters (they do not advance); its other writes land in per-call state the next call resets.
constexpr int GATE_MAX_ENVS = 1024;
// (block / n_blocks: the 256-thread blocks that share the work -- the launch's own grid, or 0 / 1 when the LAST block of
//  episode_finish_gate_kernel carries the gate)
__device__ __forceinline__ void marl_loop_gate_body(const xrl_marl_gate_t& g, int block, int n_blocks) {
    // the step's remaining per-row bookkeeping rides along: reset flags of the rows whose env finished (a dry step's are
    // wiped by the next call), RNG step counters advanced by `active`
    const int t = block * 256 + (int)threadIdx.x;
    if (g.reset_rows)
        for (int q = t; q < g.n_envs * g.n_agents; q += n_blocks * 256)
            g.reset_rows[q] = g.reset_rule ? (q < g.n_envs ? g.done[q] : 0.f) : g.done[q / g.n_agents];
    if (g.stored_state && threadIdx.x >= 64) {                            // xrl_marl_stored_state for the next step, riding along
        // Waves 1..3 of every block do this; wave 0 -- block 0's carries the loop bookkeeping, this launch's chain -- goes straight on.
        // Every thread requests ITS elements of the next state before anybody knows whether an env finished: in the common step (no
        // episode end) the copy costs no dependent round trip; only a step that did end an episode reads the finished env's row
        // behind the scan.  No workgroup barrier: each wave finds the last finished env for itself.
        constexpr int PRE = 16;
        const int tot = g.n_envs * g.state_dim, nthr = n_blocks * 192, tt = block * 192 + ((int)threadIdx.x - 64);
        const bool pre = tot <= nthr * PRE;
        float own[PRE];
        if (pre) {
#pragma unroll
            for (int k = 0; k < PRE; ++k) { const int i = tt + k * nthr; own[k] = i < tot ? g.next_state[i] : 0.f; }
        }
        int last = -1;
        for (int e = threadIdx.x & 63; e < g.n_envs; e += 64) if (g.done[e] != 0.f) last = e;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
        if (last < 0 && pre) {
#pragma unroll
            for (int k = 0; k < PRE; ++k) { const int i = tt + k * nthr; if (i < tot) g.stored_state[i] = own[k]; }
        } else {
            for (int i = tt; i < tot; i += nthr) {
                const int e = i / g.state_dim;
                g.stored_state[i] = g.next_state[(size_t)(last >= 0 ? last : e) * g.state_dim + (i - e * g.state_dim)];
            }
        }
    }
    if (t >= 64) return;
    // The first wave counts the finished envs together (one load per lane instead of one thread walking n_envs dependent loads),
    // and everything lane 0 reads below is requested here, unconditionally, in one go: this launch sits on the critical path of
    // every vector step and used to be a chain of ~8 memory round trips on a single thread.
    // (end_step: the finished envs' episode lengths go through LDS in env order for lane 0's per-env epsilon updates)
    __shared__ int s_len[GATE_MAX_ENVS];
    int c = 0;
    if (g.ptr_size || g.end_step) {
        for (int j = t; j < g.n_envs; j += 64) {
            const bool d = g.done[j] != 0.f;
            c += d;
            if (g.end_step && j < GATE_MAX_ENVS) s_len[j] = d ? g.end_step[j] : 0;
        }
        c = wave_sum(c);
    }
    if (t != 0) return;                                                   // (s_len: written and read by this one wave)
    int act = *g.active;                                                  // did the step that just ran count?
    const long long tot0 = g.totals[0], tot1 = g.totals[1], bas0 = g.base[0], bas1 = g.base[1], call0 = g.call[0], call1 = g.call[1];
    double e = *g.e_state;
    const int ps0 = g.ptr_size ? g.ptr_size[0] : 0, ps1 = g.ptr_size ? g.ptr_size[1] : 0;
    const int seq = g.host_flags ? *g.seq : 0;
    if (g.counters) { g.counters[0] += (unsigned)act; g.counters[1] += (unsigned)act; }
    if (g.ptr_size && act) {                                              // episode_advance_kernel's statements
        g.ptr_size[0] = (ps0 + c) % g.buffer_size;
        const int sz = ps1 + c;
        g.ptr_size[1] = sz < g.buffer_size ? sz : g.buffer_size;
    }
    if (act) {
        const long long ep = tot0 - bas0, st = tot1 - bas1, st_prev = g.snap[1];
        g.snap[0] = ep; g.snap[1] = st;
        if (g.end_step && c >= 2) {                                       // per finished env, in env order (:532-534)
            long long cur_i = call0 + st_prev;
            for (int j = 0; j < g.n_envs; ++j) {
                // (envs beyond the LDS copy are read from memory -- the rule is the same for any n_envs, as in the eager loop)
                const int len = j < GATE_MAX_ENVS ? s_len[j] : (g.done[j] != 0.f ? g.end_step[j] : 0);
                if (len > 0) {
                    cur_i += len;
                    e = (e > g.end_greedy) ? g.start_greedy - g.delta_greedy * (double)cur_i : g.end_greedy;
                }
            }
        } else if (!g.end_step || c >= 1) {
            const double cur = (double)(call0 + st);
            e = (e > g.end_greedy) ? g.start_greedy - g.delta_greedy * cur : g.end_greedy;
        }
        *g.e_state = e;
        *g.eps_dev = (float)e;
        if (ep >= call1) act = 0;
        *g.active = act;
    }
    *g.active_f = act ? 1.f : 0.f;
    if (g.active_i) { g.active_i[0] = act; g.active_i[1] = act; }
    if (g.host_flags) {                                                   // the host's copy: slot (launch index mod ring)
        const int k = seq;
        *g.seq = k + 1;
        __hip_atomic_store(g.host_flags + (k % g.ring), act + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // 1 / 2; the host keeps 0 = "not yet"
    }
}

__global__ void marl_loop_gate_kernel(xrl_marl_gate_t g) { marl_loop_gate_body(g, blockIdx.x, gridDim.x); }

// xrl_episode_store_finish and xrl_marl_loop_gate as ONE launch (round 6): the captured vector step of the recurrent agents was
// act / provider / store + finish / gate, four launches of 5-6 us for ~23 us of work; the gate only has to run after every block of the
// finish has read what it rewrites (active_f, the ring's ptr).  Each block takes a ticket when it is done; the block that draws the last
// one carries the gate's statements (one 256-thread block does what the gate's own launch did with one block at these sizes) and
// re-arms the ticket.  No block waits for another.
__global__ void __launch_bounds__(256) episode_finish_gate_kernel(EpPack f, const float* __restrict__ gate, const float* __restrict__ done,
                                                                  const int32_t* __restrict__ end_step, const int32_t* __restrict__ ptr_size,
                                                                  int buffer_size, const int32_t* __restrict__ steps, xrl_marl_gate_t g,
                                                                  unsigned int* __restrict__ ticket) {
    episode_finish_body(f, gate, done, end_step, ptr_size, buffer_size, steps);
    __shared__ int s_last;
    __syncthreads();
    // (no fences: the gate reads nothing a finish block writes; what it WRITES -- active_f, the ring's ptr -- the other blocks have READ,
    //  and a load whose value has come back (the __syncthreads above waits for this block's) cannot see a later store.  An agent-scope
    //  release per block -- 576 L2 write-backs per launch -- made the launch 30 us)
    if (threadIdx.x == 0) {
        const unsigned total = gridDim.x * gridDim.y;
        const unsigned k = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = k == total - 1u;
        if (s_last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    marl_loop_gate_body(g, 0, 1);
}

static int pack(const xrl_episode_field_t* fields, int n, EpPack& p) {
    if (!fields || n <= 0 || n > EP_MAX_FIELDS) return XRL_EINVAL;
    p.n = n;
    for (int i = 0; i < n; ++i) {
        if (!fields[i].a || !fields[i].b || fields[i].row_bytes <= 0 || (fields[i].row_bytes & 3) || fields[i].slots <= 0)
            return XRL_EINVAL;
        p.a[i] = fields[i].a; p.b[i] = fields[i].b; p.c[i] = fields[i].c; p.d[i] = fields[i].d;
        p.row_bytes[i] = fields[i].row_bytes; p.slots[i] = fields[i].slots; p.flags[i] = fields[i].flags;
    }
    return XRL_OK;
}

}  // namespace xrl

using namespace xrl;

extern "C" int xrl_episode_store_step(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, int n_envs,
                                      xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(steps && n_envs > 0);
    hipLaunchKernelGGL(episode_store_kernel, dim3(n_envs, n_fields), dim3(64), 0, as_stream(stream), p, steps);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_finish_gated(const xrl_episode_field_t* fields, int n_fields, const float* gate, const float* done,
                                        const int32_t* end_step, int32_t* ptr_size, int n_envs, int buffer_size, int advance,
                                        xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(done && end_step && ptr_size && n_envs > 0 && buffer_size > 0);
    hipLaunchKernelGGL(episode_finish_kernel, dim3(n_envs, n_fields), dim3(256), 0, as_stream(stream), p, gate, done, end_step,
                       ptr_size, n_envs, buffer_size, (const int32_t*)nullptr);
    if (advance)
        hipLaunchKernelGGL(episode_advance_kernel, dim3(1), dim3(1), 0, as_stream(stream), gate, done, ptr_size, n_envs,
                           buffer_size);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_store_finish(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, const float* gate,
                                        const float* done, const int32_t* end_step, const int32_t* ptr_size, int n_envs,
                                        int buffer_size, xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(steps && done && end_step && ptr_size && n_envs > 0 && buffer_size > 0);
    hipLaunchKernelGGL(episode_finish_kernel, dim3(n_envs, n_fields), dim3(256), 0, as_stream(stream), p, gate, done, end_step,
                       ptr_size, n_envs, buffer_size, steps);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_store_finish_gate(const xrl_episode_field_t* fields, int n_fields, const int32_t* steps, const float* gate,
                                             const float* done, const int32_t* end_step, const int32_t* ptr_size, int n_envs,
                                             int buffer_size, const xrl_marl_gate_t* loop_gate, uint32_t* ticket, xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(steps && done && end_step && ptr_size && n_envs > 0 && buffer_size > 0 && loop_gate && ticket);
    const xrl_marl_gate_t& g = *loop_gate;
    XRL_CHECK_ARG(g.totals && g.base && g.snap && g.active && g.call && g.e_state && g.eps_dev && g.active_f);
    XRL_CHECK_ARG(g.host_flags == nullptr || (g.seq != nullptr && g.ring > 0));
    XRL_CHECK_ARG(g.reset_rows == nullptr || (g.done && g.n_envs > 0 && g.n_agents > 0));
    XRL_CHECK_ARG(g.ptr_size == nullptr || (g.done && g.n_envs > 0 && g.buffer_size > 0));
    XRL_CHECK_ARG(g.stored_state == nullptr || (g.next_state && g.done && g.n_envs > 0 && g.state_dim > 0 && g.next_state != g.stored_state));
    hipLaunchKernelGGL(episode_finish_gate_kernel, dim3(n_envs, n_fields), dim3(256), 0, as_stream(stream), p, gate, done, end_step,
                       ptr_size, buffer_size, steps, g, ticket);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_finish(const xrl_episode_field_t* fields, int n_fields, const float* done,
                                  const int32_t* end_step, int32_t* ptr_size, int n_envs, int buffer_size,
                                  xrl_stream_t stream) {
    return xrl_episode_finish_gated(fields, n_fields, nullptr, done, end_step, ptr_size, n_envs, buffer_size, 1, stream);
}

extern "C" int xrl_host_device_pointer(void* pinned_host, void** device_out) {
    XRL_CHECK_ARG(pinned_host != nullptr && device_out != nullptr);
    XRL_CHECK_HIP(hipHostGetDevicePointer(device_out, pinned_host, 0));
    return XRL_OK;
}

namespace xrl {
// The global state the reference's multi-agent loops STORE for a vector step that follows an episode end: every env's row is
// the reset state of the LAST env that finished in the previous step (off_policy_marl.py:395 and :507 assign
// `state = info[i]["reset_state"]` -- the whole list, not `state[i]` -- and store_experience hands that single vector to the
// buffer, whose per-env write broadcasts it, memory_tools_marl.py:731-740); without an episode end, or with zero_done, a copy.
// state rows of finished envs already hold their reset state (the providers' auto-reset contract).  One workgroup per env row.
__global__ void __launch_bounds__(64) marl_stored_state_kernel(const float* state, const float* done, float* out, int n, int S) {
    __shared__ int last;
    if (threadIdx.x == 0) last = -1;
    __syncthreads();
    if (done) {
        int l = -1;
        for (int e = threadIdx.x; e < n; e += blockDim.x) if (done[e] != 0.f) l = e;
        if (l >= 0) atomicMax(&last, l);
    }
    __syncthreads();
    const int src = last >= 0 ? last : (int)blockIdx.x;
    for (int j = threadIdx.x; j < S; j += blockDim.x) out[(size_t)blockIdx.x * S + j] = state[(size_t)src * S + j];
}
}  // namespace xrl

extern "C" int xrl_marl_stored_state(const float* state, const float* done_prev, float* out, int n_envs, int state_dim,
                                     xrl_stream_t stream) {
    XRL_CHECK_ARG(state && out && state != out && n_envs > 0 && state_dim > 0);
    hipLaunchKernelGGL(xrl::marl_stored_state_kernel, dim3(n_envs), dim3(64), 0, as_stream(stream), state, done_prev, out, n_envs, state_dim);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_marl_loop_gate(const xrl_marl_gate_t* gate, xrl_stream_t stream) {
    XRL_CHECK_ARG(gate != nullptr);
    const xrl_marl_gate_t& g = *gate;
    XRL_CHECK_ARG(g.totals && g.base && g.snap && g.active && g.call && g.e_state && g.eps_dev && g.active_f);
    XRL_CHECK_ARG(g.host_flags == nullptr || (g.seq != nullptr && g.ring > 0));
    XRL_CHECK_ARG(g.reset_rows == nullptr || (g.done && g.n_envs > 0 && g.n_agents > 0));
    XRL_CHECK_ARG(g.ptr_size == nullptr || (g.done && g.n_envs > 0 && g.buffer_size > 0));
    XRL_CHECK_ARG(g.stored_state == nullptr || (g.next_state && g.done && g.n_envs > 0 && g.state_dim > 0 && g.next_state != g.stored_state));
    const int n = g.reset_rows ? g.n_envs * g.n_agents : 1;
    hipLaunchKernelGGL(marl_loop_gate_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), g);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_gather(const xrl_episode_field_t* fields, int n_fields, const int64_t* idx, int B,
                                  xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(idx && B > 0);
    hipLaunchKernelGGL(episode_gather_kernel, dim3(B, n_fields), dim3(256), 0, as_stream(stream), p, idx, B);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_episode_gather_sampled(const xrl_episode_field_t* fields, int n_fields, int64_t* idx_out, int B, int n_size,
                                          const int32_t* size_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_dev,
                                          xrl_stream_t stream) {
    EpPack p;
    XRL_CHECK_ARG(pack(fields, n_fields, p) == XRL_OK);
    XRL_CHECK_ARG(idx_out && size_dev && B > 0 && n_size > 0);
    hipLaunchKernelGGL(episode_gather_sampled_kernel, dim3(B, n_fields), dim3(256), 0, as_stream(stream), p, idx_out, B, n_size, size_dev,
                       seed, counter, counter_dev);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
