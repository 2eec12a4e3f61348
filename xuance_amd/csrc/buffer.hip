// Rollout / replay buffer kernels: time-major structure-of-arrays store, gather, GAE scan, adv statistics.
// Reference arithmetic: xuance/common/memory_tools.py (store_element :44-61, finish_path :242-265,
// sample :267-287 / :374-387).  All HBM-bound; layouts are chosen so every access is coalesced.
#include "common.h"
#include "rng.h"
#include <stdarg.h>

namespace xrl {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

constexpr int MAX_FIELDS = 16;
struct FieldPack {
    void* dst[MAX_FIELDS];
    const void* src[MAX_FIELDS];
    int row_bytes[MAX_FIELDS];
    int flags[MAX_FIELDS];
    int n;
};

// ------------------------------------------------------------------------------------------ store
// field[t] <- src : one contiguous copy of n_envs*row_bytes per field (blockIdx.y = field).
// RING: slot and filled-slot count from a device counter (xrl_soa_store_step_ring); every block reads the counter, nobody writes it
struct RingCursor { const int32_t* counter; int32_t offset, n_size; long long slot_bias, size_bias; };
template <typename V>
__global__ void __launch_bounds__(256) store_step_kernel(FieldPack f, int n_envs, int t, int32_t* size_dev, int32_t new_size, RingCursor rc) {
    const int fi = blockIdx.y;
    if (rc.counter) {
        const long long c = (long long)*rc.counter + rc.offset;
        long long sl = (rc.slot_bias + c) % rc.n_size;
        if (sl < 0) sl += rc.n_size;
        t = (int)sl;
        const long long ns = rc.size_bias + c + 1;
        new_size = (int32_t)(ns < rc.n_size ? ns : rc.n_size);
    }
    if (size_dev && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *size_dev = new_size;   // filled slots of a replay ring
    const size_t step_bytes = (size_t)n_envs * f.row_bytes[fi];
    const size_t n = step_bytes / sizeof(V);
    const V* __restrict__ s = reinterpret_cast<const V*>(f.src[fi]);
    V* __restrict__ d = reinterpret_cast<V*>(reinterpret_cast<char*>(f.dst[fi]) + (size_t)t * step_bytes);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

// ------------------------------------------------------------------------------------------ gather
// dst[b] = field[t_b][env_b], (env_b, t_b) = divmod(idx[b], T).  Unit V = 16 B or 4 B.
// SAMPLED: the indices are not read but drawn here (bs <= 256: every workgroup draws all of them into LDS; workgroup
// (0, 0) also writes them to draw.idx_out) -- one launch instead of xrl_sample_replay_indices + xrl_soa_gather.
template <typename V, bool SAMPLED>
__global__ void __launch_bounds__(256) gather_kernel(FieldPack f, const int64_t* __restrict__ idx, int bs,
                                                     int n_envs, int T, const float* __restrict__ stats, ReplayDraw draw) {
    __shared__ int64_t sidx[SAMPLED ? 256 : 1];
    if (SAMPLED) {
        if ((int)threadIdx.x < bs) {
            const int64_t v = replay_draw(draw, threadIdx.x, n_envs, T);
            sidx[threadIdx.x] = v;
            if (blockIdx.x == 0 && blockIdx.y == 0 && draw.idx_out) draw.idx_out[threadIdx.x] = v;
        }
        __syncthreads();
    }
    const int fi = blockIdx.y;
    if constexpr (sizeof(V) == 4) {
        // a field whose rows and bases allow 16-byte units takes them even when another field of the launch does not (round 6: one
        // 4-byte field next to the 28 KB frame rows of PPO on frame stacks made the whole launch copy in 4-byte units, 10.3 us)
        if (f.flags[fi] & 2) {
            const int rw16 = f.row_bytes[fi] / 16;
            const size_t total16 = (size_t)bs * rw16;
            const uint4* __restrict__ s16 = reinterpret_cast<const uint4*>(f.src[fi]);
            uint4* __restrict__ d16 = reinterpret_cast<uint4*>(f.dst[fi]);
            for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total16; i += (size_t)gridDim.x * blockDim.x) {
                const int b = (int)(i / rw16), w = (int)(i - (size_t)b * rw16);
                const int64_t fl = SAMPLED ? sidx[b] : idx[b];
                const int env = (int)(fl / T), t = (int)(fl - (int64_t)env * T);
                d16[i] = s16[((size_t)t * n_envs + env) * rw16 + w];
            }
            return;
        }
    }
    const int rw = f.row_bytes[fi] / (int)sizeof(V);  // units per row
    const size_t total = (size_t)bs * rw;
    const V* __restrict__ s = reinterpret_cast<const V*>(f.src[fi]);
    V* __restrict__ d = reinterpret_cast<V*>(f.dst[fi]);
    const bool norm = (f.flags[fi] & 1) && stats != nullptr;
    float mean = 0.f, denom = 1.f;
    if (norm) {
        mean = stats[0];
        denom = stats[1] + 1e-8f;  // np.std(adv) + 1e-8   (memory_tools.py:282)
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / rw), w = (int)(i - (size_t)b * rw);
        const int64_t fl = SAMPLED ? sidx[b] : idx[b];
        const int env = (int)(fl / T), t = (int)(fl - (int64_t)env * T);
        V v = s[((size_t)t * n_envs + env) * rw + w];
        if constexpr (sizeof(V) == 4) {
            if (norm) {
                float x = __builtin_bit_cast(float, v);
                x = __fdiv_rn(__fsub_rn(x, mean), denom);
                v = __builtin_bit_cast(V, x);
            }
        }
        d[i] = v;
    }
}

// ------------------------------------------------------------------------------------------ adv stats
// One block per minibatch: two-pass mean / population std in float64 over adv[idx[...]].
__global__ void __launch_bounds__(1024) adv_stats_kernel(const float* __restrict__ adv, const int64_t* __restrict__ idx,
                                                         int bs, int n_envs, int T, float* __restrict__ stats) {
    __shared__ double scratch[16];
    const int64_t* my = idx + (size_t)blockIdx.x * bs;
    // (round 6: a thread's first eight indices, then its eight values, are requested together and the values kept for the second pass;
    //  per-thread sums run in the same order as the plain loops did -- 32 dependent round trips per launch before, 15.8 us)
    constexpr int NK = 8;
    const int nth = blockDim.x;
    float keep[NK];
    {
        int64_t fl[NK];
#pragma unroll
        for (int u = 0; u < NK; ++u) { const int i = threadIdx.x + u * nth; fl[u] = i < bs ? my[i] : 0; }
#pragma unroll
        for (int u = 0; u < NK; ++u) {
            const int env = (int)(fl[u] / T), t = (int)(fl[u] - (int64_t)env * T);
            keep[u] = threadIdx.x + u * nth < bs ? adv[(size_t)t * n_envs + env] : 0.f;
        }
    }
    double s = 0.0;
#pragma unroll
    for (int u = 0; u < NK; ++u) if (threadIdx.x + u * nth < bs) s += (double)keep[u];
    for (int i = threadIdx.x + NK * nth; i < bs; i += nth) {
        const int64_t fl = my[i];
        const int env = (int)(fl / T), t = (int)(fl - (int64_t)env * T);
        s += (double)adv[(size_t)t * n_envs + env];
    }
    const double mean = block_sum(s, scratch) / bs;
    double q = 0.0;
#pragma unroll
    for (int u = 0; u < NK; ++u) if (threadIdx.x + u * nth < bs) { const double dlt = (double)keep[u] - mean; q += dlt * dlt; }
    for (int i = threadIdx.x + NK * nth; i < bs; i += nth) {
        const int64_t fl = my[i];
        const int env = (int)(fl / T), t = (int)(fl - (int64_t)env * T);
        const double dlt = (double)adv[(size_t)t * n_envs + env] - mean;
        q += dlt * dlt;
    }
    const double var = block_sum(q, scratch) / bs;
    if (threadIdx.x == 0) {
        stats[2 * blockIdx.x + 0] = (float)mean;
        stats[2 * blockIdx.x + 1] = (float)sqrt(var);
    }
}

// ------------------------------------------------------------------------------------------ GAE scan
// One lane per env, backwards in time; loads of a chunk of CH steps are issued before the recurrence so
// CH*5 independent coalesced loads are in flight per lane.  Arithmetic is written with explicit
// round-to-nearest intrinsics (no FMA contraction) in the operand order of memory_tools.py:255-257 so the
// result is bit-identical to NumPy's float32 (seg bit1 = 0) or float64-carry (bit1 = 1) evaluation.
template <int CH>
__global__ void __launch_bounds__(64) gae_scan_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                      const float* __restrict__ term, const float* __restrict__ bootv,
                                                      const uint8_t* __restrict__ seg, float* __restrict__ adv,
                                                      float* __restrict__ ret, int n_envs, int T, float gamma_f,
                                                      float lam_f, double gamma_d, int use_gae) {
#pragma clang fp contract(off)   // bit-exactness: every multiply and add rounds separately, like NumPy
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_envs) return;
    bool active = false, m64 = false, first = true;
    float last_f = 0.f, vnext_f = 0.f;
    double last_d = 0.0, vnext_d = 0.0, disc_d = 0.0;  // disc_d: discounted reward sum for use_gae == 0
    // the chunk after the current one is loaded while the current one is scanned (two chunks of 5*CH loads in flight)
    float nr[CH], nv[CH], nd_[CH], nbv[CH];
    uint8_t nsg[CH];
    auto load_chunk = [&](int t0) {
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            const int t = t0 + j;
            if (t >= 0) {
                const size_t o = (size_t)t * n_envs + e;
                nr[j] = rew[o]; nv[j] = val[o]; nd_[j] = term[o]; nbv[j] = bootv[o]; nsg[j] = seg[o];
            } else { nr[j] = nv[j] = nd_[j] = nbv[j] = 0.f; nsg[j] = 0; }
        }
    };
    load_chunk(T - CH);
    for (int t1 = T; t1 > 0; t1 -= CH) {
        const int t0 = t1 - CH;  // chunk covers t0 .. t1-1 (t0 may be negative)
        float r[CH], v[CH], d[CH], bv[CH];
        uint8_t sg[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) { r[j] = nr[j]; v[j] = nv[j]; d[j] = nd_[j]; bv[j] = (nsg[j] & 4) ? 0.f : nbv[j]; sg[j] = nsg[j]; }
        if (t0 > 0) load_chunk(t0 - CH);
#pragma unroll
        for (int j = CH - 1; j >= 0; --j) {
            const int t = t0 + j;
            if (t < 0) continue;
            if (sg[j] & 1) {  // a finish_path(val, env) call closed a path after step t
                active = true; first = true;
                m64 = (sg[j] & 2) != 0;
                last_f = 0.f; last_d = 0.0;
                vnext_f = bv[j]; vnext_d = (double)bv[j];
                disc_d = (double)bv[j];
            }
            if (!active) continue;
            const size_t o = (size_t)t * n_envs + e;
            const float nd = __fsub_rn(1.0f, d[j]);
            const float c1 = __fmul_rn(nd, gamma_f);                 // (1 - d) * gamma      (float32)
            if (use_gae) {
                const float c2 = __fmul_rn(c1, lam_f);               // (1 - d) * gamma * lam (float32)
                if (!m64) {
                    const float delta = __fsub_rn(__fadd_rn(r[j], __fmul_rn(c1, vnext_f)), v[j]);
                    last_f = __fadd_rn(delta, __fmul_rn(c2, last_f));
                    adv[o] = last_f;
                    ret[o] = __fadd_rn(last_f, v[j]);
                } else {
                    const double vd = (double)v[j];
                    const double delta = __dsub_rn(__dadd_rn((double)r[j], __dmul_rn((double)c1, vnext_d)), vd);
                    // first step of a path: `last` is the Python int 0 -> c2 * 0 stays float32
                    const double carry = first ? (double)__fmul_rn(c2, 0.0f) : __dmul_rn((double)c2, last_d);
                    last_d = __dadd_rn(delta, carry);
                    const float a = (float)last_d;
                    adv[o] = a;
                    ret[o] = (float)__dadd_rn((double)a, vd);
                }
            } else {
                // returns = discount_cumsum(rewards + [val])[:-1] (float64 filter); adv = r + gamma*v' - v
                disc_d = (double)r[j] + gamma_d * disc_d;
                ret[o] = (float)disc_d;
                if (!m64) adv[o] = __fsub_rn(__fadd_rn(r[j], __fmul_rn(gamma_f, vnext_f)), v[j]);
                else adv[o] = (float)__dsub_rn(__dadd_rn((double)r[j], __dmul_rn(gamma_d, vnext_d)), (double)v[j]);
            }
            first = false;
            vnext_f = v[j]; vnext_d = (double)v[j];
        }
    }
}

// ---- "relay" formulation used for every launch (the lane-per-env kernel above is kept for reference / tiny T) -------------
// A workgroup owns 64 envs and NWV waves; wave w owns CH consecutive timesteps of a super-chunk of CH*NWV steps.
//   phase 1 (all waves in parallel): every wave loads its CH steps (5 coalesced loads per step) and precomputes, for
//           both carry precisions, everything that does not depend on the carried advantage: c2 = (1-d)*gamma*lam and
//           delta = r + (1-d)*gamma*v_next - v  (v_next is the next step's value, the bootstrap value at a path end, or the
//           first value of the following chunk) -- ONE global round trip per super-chunk instead of one per 8 steps;
//   phase 2 (relay, latest wave first): the wave whose turn it is reads the carried state from LDS, runs the two-op
//           recurrence last = delta + c2*last over its CH steps, stores advantages/returns, hands the state on.
// The arithmetic per step is unchanged (same operations, same order, no contraction), so results stay bit-identical to NumPy.
template <int CH, int NWV>
__global__ void __launch_bounds__(64 * NWV) gae_relay_kernel(const float* __restrict__ rew, const float* __restrict__ val,
                                                             const float* __restrict__ term, const float* __restrict__ bootv,
                                                             const uint8_t* __restrict__ seg, float* __restrict__ adv,
                                                             float* __restrict__ ret, int n_envs, int T, float gamma_f,
                                                             float lam_f, double gamma_d, int use_gae) {
#pragma clang fp contract(off)
    __shared__ float c_lastf[64];
    __shared__ double c_lastd[64], c_disc[64];
    __shared__ int c_flags[64];                     // bit0 active, bit1 float64 carry
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int e = blockIdx.x * 64 + lane;
    const bool valid = e < n_envs;
    if (w == 0) { c_lastf[lane] = 0.f; c_lastd[lane] = 0.0; c_disc[lane] = 0.0; c_flags[lane] = 0; }
    __syncthreads();
    for (int hi = T; hi > 0; hi -= CH * NWV) {
        const int t0 = hi - CH * (NWV - w);         // this wave covers [t0, t0 + CH); steps < 0 do not exist
        float v[CH], c2[CH], df[CH], bvk[CH], vnk[CH];
        double dd[CH];
        uint8_t sg[CH];
        {
            float r[CH], d[CH], bv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int t = t0 + j;
                if (valid && t >= 0) {
                    const size_t o = (size_t)t * n_envs + e;
                    r[j] = rew[o]; v[j] = val[o]; d[j] = term[o]; bv[j] = bootv[o]; sg[j] = seg[o];
                    if (sg[j] & 4) bv[j] = 0.f;               // device rollouts: the env terminated, finish_path(0.0, i)
                } else { r[j] = v[j] = d[j] = bv[j] = 0.f; sg[j] = 0; }
            }
            const int tn = t0 + CH;
            const float vext = (valid && tn >= 0 && tn < T) ? val[(size_t)tn * n_envs + e] : 0.f;
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const float vn = (sg[j] & 1) ? bv[j] : (j + 1 < CH ? v[j + 1] : vext);
                const float nd = __fsub_rn(1.0f, d[j]);
                const float c1 = use_gae ? __fmul_rn(nd, gamma_f) : gamma_f;                   // (1 - d) * gamma
                c2[j] = __fmul_rn(c1, lam_f);                                                  // (1 - d) * gamma * lam
                df[j] = __fsub_rn(__fadd_rn(r[j], __fmul_rn(c1, vn)), v[j]);                   // memory_tools.py:255 / :261
                dd[j] = use_gae ? __dsub_rn(__dadd_rn((double)r[j], __dmul_rn((double)c1, (double)vn)), (double)v[j])
                                : __dsub_rn(__dadd_rn((double)r[j], __dmul_rn(gamma_d, (double)vn)), (double)v[j]);
                if (!use_gae) c2[j] = r[j];          // non-GAE: the relay needs the raw reward for the discounted sum
                bvk[j] = bv[j]; vnk[j] = vn;
            }
        }
        // phase 2: relay.  Nothing touches global memory here (results stay in registers), so the barriers are cheap.
        unsigned wmask = 0, m64mask = 0;            // per-step: result valid / float64 path
        for (int turn = NWV - 1; turn >= 0; --turn) {
            if (w == turn && valid && t0 + CH > 0) {
                int fl = c_flags[lane];
                bool active = fl & 1, m64 = (fl & 2) != 0;
                float last_f = c_lastf[lane];
                double last_d = c_lastd[lane], disc = c_disc[lane];
                if (use_gae) {
                    // branch-free: both carry precisions advance at every step (each is reset at a path end, and a path
                    // has one precision for its whole length, so the chain that is read never sees the other's steps);
                    // the per-step work is two dependent fp32 and two dependent fp64 operations instead of a divergent
                    // if / else nest (measured: ~450 cycles per step before)
#pragma unroll
                    for (int j = CH - 1; j >= 0; --j) {
                        const bool end = (sg[j] & 1) != 0;
                        active = active || end;
                        m64 = end ? ((sg[j] & 2) != 0) : m64;
                        last_f = end ? 0.f : last_f;
                        last_d = end ? 0.0 : last_d;
                        last_f = __fadd_rn(df[j], __fmul_rn(c2[j], last_f));                   // :256
                        last_d = __dadd_rn(dd[j], __dmul_rn((double)c2[j], last_d));
                        const bool live = active && (t0 + j >= 0);
                        df[j] = m64 ? (float)last_d : last_f;
                        wmask |= live ? (1u << j) : 0u;
                        m64mask |= (live && m64) ? (1u << j) : 0u;
                    }
                } else
#pragma unroll
                for (int j = CH - 1; j >= 0; --j) {
                    if (t0 + j < 0) continue;
                    if (sg[j] & 1) {                  // finish_path(val, env) closed a path after step t
                        active = true; m64 = (sg[j] & 2) != 0; last_f = 0.f; last_d = 0.0;
                        disc = (double)bvk[j];
                    }
                    if (!active) continue;
                    wmask |= 1u << j;
                    if (m64) m64mask |= 1u << j;
                    if (use_gae) {
                        if (!m64) {
                            last_f = __fadd_rn(df[j], __fmul_rn(c2[j], last_f));               // :256
                            df[j] = last_f;
                        } else {
                            last_d = __dadd_rn(dd[j], __dmul_rn((double)c2[j], last_d));
                            df[j] = (float)last_d;
                        }
                    } else {
                        disc = (double)c2[j] + gamma_d * disc;                                 // discount_cumsum, float64
                        dd[j] = disc;                                                          // -> returns
                        if (m64) df[j] = (float)__dsub_rn(__dadd_rn((double)c2[j], __dmul_rn(gamma_d, (double)vnk[j])), (double)v[j]);
                    }
                }
                c_flags[lane] = (active ? 1 : 0) | (m64 ? 2 : 0);
                c_lastf[lane] = last_f; c_lastd[lane] = last_d; c_disc[lane] = disc;
            }
            __syncthreads();
        }
        // phase 3: all waves store their results together (one store drain per super-chunk)
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (wmask & (1u << j)) {
                const size_t o = (size_t)(t0 + j) * n_envs + e;
                const float a = df[j];
                adv[o] = a;
                if (use_gae) ret[o] = (m64mask & (1u << j)) ? (float)__dadd_rn((double)a, (double)v[j]) : __fadd_rn(a, v[j]);   // :257
                else ret[o] = (float)dd[j];
            }
        }
    }
}

static int pack_fields(const xrl_field_t* fields, int n_fields, FieldPack& fp, bool& vec16) {
    if (fields == nullptr || n_fields < 1 || n_fields > MAX_FIELDS) return XRL_EINVAL;
    fp.n = n_fields;
    vec16 = true;
    for (int i = 0; i < n_fields; ++i) {
        if (!fields[i].dst || !fields[i].src || fields[i].row_bytes <= 0 || (fields[i].row_bytes & 3)) return XRL_EINVAL;
        fp.dst[i] = fields[i].dst; fp.src[i] = fields[i].src;
        fp.row_bytes[i] = fields[i].row_bytes; fp.flags[i] = fields[i].flags;
        if ((fields[i].row_bytes & 15) || ((uintptr_t)fields[i].dst & 15) || ((uintptr_t)fields[i].src & 15)) vec16 = false;
        else if (!(fields[i].flags & 1)) fp.flags[i] |= 2;              // (gather_kernel: this field alone may go in 16-byte units)
    }
    return XRL_OK;
}

}  // namespace xrl

using namespace xrl;

extern "C" const char* xrl_version(void) { return "xrl_hip 0.1.0 (gfx950)"; }
extern "C" const char* xrl_last_error(void) { return xrl::g_err; }

extern "C" int xrl_device_info(int* cu_count, int* wave_size, char* arch, int arch_len) {
    int dev = 0;
    XRL_CHECK_HIP(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    XRL_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) { strncpy(arch, prop.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    return XRL_OK;
}

static int store_step_launch(const xrl_field_t* fields, int n_fields, int n_envs, int t, int32_t* size_dev, int32_t new_size,
                             const RingCursor& rc, xrl_stream_t stream) {
    FieldPack fp; bool vec16;
    XRL_CHECK_ARG(pack_fields(fields, n_fields, fp, vec16) == XRL_OK);
    XRL_CHECK_ARG(n_envs > 0 && t >= 0);
    size_t max_bytes = 0;
    for (int i = 0; i < n_fields; ++i) {
        const size_t sb = (size_t)n_envs * fp.row_bytes[i];
        max_bytes = sb > max_bytes ? sb : max_bytes;
        if ((sb & 15) || (!rc.counter && (((size_t)t * sb) & 15))) vec16 = false;   // (a ring cursor: every slot, sb a multiple of 16)
    }
    const size_t unit = vec16 ? 16 : 4;
    size_t nb = (max_bytes / unit + 255) / 256;
    if (nb > 2048) nb = 2048;
    if (nb < 1) nb = 1;
    dim3 grid((unsigned)nb, n_fields);
    if (vec16) hipLaunchKernelGGL(store_step_kernel<uint4>, grid, 256, 0, as_stream(stream), fp, n_envs, t, size_dev, new_size, rc);
    else hipLaunchKernelGGL(store_step_kernel<uint32_t>, grid, 256, 0, as_stream(stream), fp, n_envs, t, size_dev, new_size, rc);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_soa_store_step_sized(const xrl_field_t* fields, int n_fields, int n_envs, int t, int32_t* size_dev, int32_t new_size,
                                        xrl_stream_t stream) {
    return store_step_launch(fields, n_fields, n_envs, t, size_dev, new_size, RingCursor{nullptr, 0, 1, 0, 0}, stream);
}

extern "C" int xrl_soa_store_step_ring(const xrl_field_t* fields, int n_fields, int n_envs, int n_size, int64_t slot_bias, int64_t size_bias,
                                       const int32_t* counter_dev, int32_t offset, int32_t* size_dev, xrl_stream_t stream) {
    XRL_CHECK_ARG(counter_dev != nullptr && n_size >= 1);
    return store_step_launch(fields, n_fields, n_envs, 0, size_dev, 0, RingCursor{counter_dev, offset, n_size, (long long)slot_bias, (long long)size_bias}, stream);
}

extern "C" int xrl_soa_store_step(const xrl_field_t* fields, int n_fields, int n_envs, int t, xrl_stream_t stream) {
    return xrl_soa_store_step_sized(fields, n_fields, n_envs, t, nullptr, 0, stream);
}

extern "C" int xrl_soa_gather(const xrl_field_t* fields, int n_fields, const int64_t* idx, int bs, int n_envs, int T,
                              const float* stats, xrl_stream_t stream) {
    FieldPack fp; bool vec16;
    XRL_CHECK_ARG(pack_fields(fields, n_fields, fp, vec16) == XRL_OK);
    XRL_CHECK_ARG(idx != nullptr && bs > 0 && n_envs > 0 && T > 0);
    for (int i = 0; i < n_fields; ++i)
        if ((fp.flags[i] & 1)) { XRL_CHECK_ARG(fp.row_bytes[i] == 4); vec16 = false; }
    size_t max_units = 0;
    for (int i = 0; i < n_fields; ++i) {
        const size_t u = (size_t)bs * fp.row_bytes[i] / (vec16 ? 16 : 4);
        max_units = u > max_units ? u : max_units;
    }
    size_t nb = (max_units + 255) / 256;
    if (nb > 4096) nb = 4096;
    dim3 grid((unsigned)nb, n_fields);
    if (vec16) hipLaunchKernelGGL((gather_kernel<uint4, false>), grid, 256, 0, as_stream(stream), fp, idx, bs, n_envs, T, stats, ReplayDraw{});
    else hipLaunchKernelGGL((gather_kernel<uint32_t, false>), grid, 256, 0, as_stream(stream), fp, idx, bs, n_envs, T, stats, ReplayDraw{});
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_soa_gather_sampled(const xrl_field_t* fields, int n_fields, int64_t* idx_out, int bs, int n_envs, int n_size,
                                      const int32_t* size_dev, uint64_t seed, uint32_t counter, const uint32_t* counter_dev,
                                      xrl_stream_t stream) {
    FieldPack fp; bool vec16;
    XRL_CHECK_ARG(pack_fields(fields, n_fields, fp, vec16) == XRL_OK);
    XRL_CHECK_ARG(size_dev != nullptr && bs > 0 && bs <= 256 && n_envs > 0 && n_size > 0);
    for (int i = 0; i < n_fields; ++i) XRL_CHECK_ARG(!(fp.flags[i] & 1));
    size_t max_units = 0;
    for (int i = 0; i < n_fields; ++i) {
        const size_t u = (size_t)bs * fp.row_bytes[i] / (vec16 ? 16 : 4);
        max_units = u > max_units ? u : max_units;
    }
    size_t nb = (max_units + 255) / 256;
    if (nb > 4096) nb = 4096;
    dim3 grid((unsigned)nb, n_fields);
    const ReplayDraw d{size_dev, seed, counter, counter_dev, idx_out};
    if (vec16) hipLaunchKernelGGL((gather_kernel<uint4, true>), grid, 256, 0, as_stream(stream), fp, nullptr, bs, n_envs, n_size, nullptr, d);
    else hipLaunchKernelGGL((gather_kernel<uint32_t, true>), grid, 256, 0, as_stream(stream), fp, nullptr, bs, n_envs, n_size, nullptr, d);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_adv_stats(const float* adv_field, const int64_t* idx, int bs, int n_batches, int n_envs, int T,
                             float* stats, xrl_stream_t stream) {
    XRL_CHECK_ARG(adv_field && idx && stats && bs > 0 && n_batches > 0 && n_envs > 0 && T > 0);
    hipLaunchKernelGGL(adv_stats_kernel, dim3(n_batches), dim3(1024), 0, as_stream(stream), adv_field, idx, bs, n_envs,
                       T, stats);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}

extern "C" int xrl_gae_scan(const float* rew, const float* val, const float* term, const float* bootv,
                            const uint8_t* seg, float* adv, float* ret, int n_envs, int T, double gamma, double lam,
                            int use_gae, xrl_stream_t stream) {
    XRL_CHECK_ARG(rew && val && term && bootv && seg && adv && ret && n_envs > 0 && T > 0);
    const int nb = (n_envs + 63) / 64;
    // few envs -> few workgroups: split the time axis over 8 waves per workgroup (relay); many envs: the lane-per-env
    // scan already fills the chip and streams at a higher rate (tools/scale_sweep.py)
    if (T >= 32 && n_envs <= 16384)
        hipLaunchKernelGGL((gae_relay_kernel<16, 8>), dim3(nb), dim3(512), 0, as_stream(stream), rew, val, term, bootv, seg,
                           adv, ret, n_envs, T, (float)gamma, (float)lam, gamma, use_gae);
    else
        hipLaunchKernelGGL(gae_scan_kernel<8>, dim3(nb), dim3(64), 0, as_stream(stream), rew, val, term, bootv, seg, adv,
                           ret, n_envs, T, (float)gamma, (float)lam, gamma, use_gae);
    XRL_CHECK_LAUNCH();
    return XRL_OK;
}
