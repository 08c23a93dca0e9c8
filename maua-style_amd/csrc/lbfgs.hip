// Device-resident L-BFGS for the pixel update (torch.optim.LBFGS as configured at reference optim.py:180-191:
// lr 1, no line search, history 100, tolerances -1).
//
// The two-loop recursion of torch/optim/lbfgs.py walks the history sequentially: ~4m dependent dot/axpy launches and
// 4-5 host synchronisations per iteration.  Here the same recursion is evaluated in the coefficient space of the
// stored vectors ("vector-free" form): with the basis B = [s_0..s_m, y_0..y_m, g] and its Gram matrix M = B^T B,
// every dot product of the recursion is a row of M times the coefficient vector, so one iteration needs
//   1. the new pair y = g - g_prev, s = t*d formed exactly as the reference does (fp32, elementwise)   [lbfgs_pair]
//      and ONE sweep over the history slab computing the dots of every stored vector with (s, y, g)  [lbfgs_pair_dots]
//   2. a fixed-order reduction of the per-workgroup partials                                        [lbfgs_finish_dots]
//   3. the recursion on (2m+3) coefficients in fp64 by one wave, incl. the y.s > 1e-10 test, the ring
//      update, H_diag = ys/yy, t, g.d and the stop test g.d > -tolerance_change                     [lbfgs_coeffs]
//   4. ONE more sweep: d = B * coeff, x += t*d                                                      [lbfgs_combine]
// = 2 passes over the slab (the algorithmic 4m*n*4 bytes), 5 launches, no host sync, no float atomics.
#include <stdint.h>
#include <stdlib.h>

#include "common.hpp"

// History reads of the two sweeps: 2.5 GB per sweep at 1024 x 1024, each byte used once per sweep - marked non-temporal (-DLB_NT=0: plain
// loads).  Same box, it/s: 1024x1024 188.7 -> 191.8, 512x512 558.4 -> 562.3, 256x256 1090.8 -> 1111.1.
#ifndef LB_NT
#define LB_NT 1
#endif
typedef float lb_f32x4 __attribute__((ext_vector_type(4)));
#if LB_NT
__device__ __forceinline__ float lb_stream_load(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float4 lb_stream_load(const float4* p) {
    const lb_f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const lb_f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
#define LB_STREAM_LOAD(p) lb_stream_load(p)
#else
#define LB_STREAM_LOAD(p) (*(p))
#endif

namespace maua {

struct LbfgsHeader {
    int n_iter;     // completed calls of iterate
    int len;        // pairs in the ring
    int head;       // physical slot of the oldest pair
    int stopped;    // raised by the g.d test; iterate() becomes a no-op for x
    int cand;       // physical slot the next pair is written to (always free: the ring has history+1 slots)
    int pad0;
    float t;        // step length of the LAST move (s = t*d)
    float gtd;
    double h_diag;
    float prev_loss;      // loss handed to the previous call (lbfgs.py: prev_loss)
    unsigned gmax_bits;   // max |g| of the gradient handed to this call (bit pattern of a non-negative float: orders like an integer)
    unsigned dmax_bits;   // max |t d| of the last move
    int pad1;
    double reserved[2];
};

constexpr int LB_EPT = 16;            // elements per thread in the sweeps
constexpr int LB_WG = 256 * LB_EPT;   // elements per workgroup

struct LbfgsLayout {
    int m1, nb_ids, nwg;
    size_t off_coef, off_dots, off_M, off_partial, off_gprev, off_d, off_S, off_Y, total;
};

static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

static LbfgsLayout lbfgs_layout(int64_t count, int history) {
    LbfgsLayout L;
    L.m1 = history + 1;
    L.nb_ids = 2 * L.m1 + 1;
    L.nwg = (int)((count + LB_WG - 1) / LB_WG);
    size_t o = align256(sizeof(LbfgsHeader));
    L.off_coef = o;
    o = align256(o + sizeof(float) * L.nb_ids);
    L.off_dots = o;
    o = align256(o + sizeof(double) * 4 * L.nb_ids);
    L.off_M = o;
    o = align256(o + sizeof(double) * (size_t)L.nb_ids * L.nb_ids);
    L.off_partial = o;
    o = align256(o + sizeof(float) * 4 * (size_t)L.nb_ids * L.nwg);
    L.off_gprev = o;
    o = align256(o + sizeof(float) * count);
    L.off_d = o;
    o = align256(o + sizeof(float) * count);
    L.off_S = o;
    o = align256(o + sizeof(float) * count * L.m1);
    L.off_Y = o;
    o = align256(o + sizeof(float) * count * L.m1);
    L.total = o;
    return L;
}

__global__ void lbfgs_init_kernel(LbfgsHeader* h, double* M, int nb_ids, float* coef) {
    const int64_t tot = (int64_t)nb_ids * nb_ids;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (int64_t)gridDim.x * blockDim.x) M[i] = 0.0;
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < nb_ids; i += blockDim.x) coef[i] = 0.f;
        if (threadIdx.x == 0) {
            h->n_iter = 0;
            h->len = 0;
            h->head = 0;
            h->stopped = 0;
            h->cand = 0;
            h->pad0 = 0;
            h->t = 0.f;
            h->gtd = 0.f;
            h->h_diag = 1.0;
            h->prev_loss = 0.f;
            h->gmax_bits = 0u;
            h->dmax_bits = 0u;
        }
    }
}

__device__ __forceinline__ void wave_reduce3_store(float a, float b, float c, float* dst, int lane) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
    }
    if (lane == 0) {
        dst[0] = a;
        dst[1] = b;
        dst[2] = c;
    }
}

// Pair update (elementwise): y = g - g_prev and s = t*d go to the candidate slot, g_prev = g.  First call: zeros.
__global__ void __launch_bounds__(256)
lbfgs_pair_kernel(const LbfgsHeader* __restrict__ hdr, const float* __restrict__ g, float* __restrict__ g_prev,
                  const float* __restrict__ d, float* __restrict__ S, float* __restrict__ Y, int64_t n) {
    const bool first = hdr->n_iter == 0;
    const float t = hdr->t;
    float* sc = S + (int64_t)hdr->cand * n;
    float* yc = Y + (int64_t)hdr->cand * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x) {
        const float gv = g[e];
        sc[e] = first ? 0.f : d[e] * t;          // s = d.mul(t)
        yc[e] = first ? 0.f : gv - g_prev[e];    // y = flat_grad.sub(prev_flat_grad)
        g_prev[e] = gv;
    }
}

// Sweep 1.  ids: s-slot p -> p, y-slot p -> m1 + p, g -> 2*m1.  partial[wg][id][4] = per-workgroup dots of vector id
// with (s_new, y_new, g) and, for id == g only, sum|g| in the 4th slot.  The candidate pair is read back like any
// stored pair, so the loop body is uniform: 2*LB_EPT coalesced loads, 6*LB_EPT FMAs, six 64-lane reductions.
//
// Small vectors (a 256 x 256 image is 48 blocks) leave most of the chip idle, so the pair loop is also cut into gridDim.y
// groups: workgroup (block, group) handles the pairs [group * per, (group + 1) * per) of its block.  Every (block, id) entry is
// still computed by exactly one workgroup with the same arithmetic, whatever the number of groups: bit-identical results.
// VEC: the thread's 16 elements are four float4 (count % 4 == 0, 16-byte aligned vectors) instead of 16 strided floats: a quarter
// of the memory requests for the same bytes.
template <bool VEC>
__global__ void __launch_bounds__(256)
lbfgs_pair_dots_kernel(LbfgsHeader* __restrict__ hdr, const float* __restrict__ g, const float* __restrict__ S,
                       const float* __restrict__ Y, float* __restrict__ partial, int64_t n, int m1) {
    extern __shared__ float lds[];  // [4 waves][nb_ids][4]
    const int nb_ids = 2 * m1 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cand = hdr->cand, len = hdr->len, head = hdr->head;
    const int per = (len + 1 + (int)gridDim.y - 1) / (int)gridDim.y;
    const int i_begin = (int)blockIdx.y * per, i_end = min(len + 1, i_begin + per);
    if (i_begin >= i_end && blockIdx.y != 0) return;  // (while the history fills, the last groups have nothing to do)
    for (int i = tid; i < 4 * nb_ids * 4; i += 256) lds[i] = 0.f;
    __syncthreads();
    float* mine = lds + (size_t)wave * nb_ids * 4;

    const int64_t blk = (int64_t)blockIdx.x * LB_WG;
    const int rem = (int)min((int64_t)LB_WG, n - blk);  // valid elements of this block
    int off[LB_EPT];
    float gv[LB_EPT], sv[LB_EPT], yv[LB_EPT];
    {
        const float* gB = g + blk;
        const float* sB = S + (int64_t)cand * n + blk;
        const float* yB = Y + (int64_t)cand * n + blk;
#pragma unroll
        for (int k = 0; k < LB_EPT; ++k) {
            const int e = VEC ? 4 * (tid + 256 * (k >> 2)) + (k & 3) : tid + 256 * k;
            const bool ok = e < rem;
            off[k] = ok ? e : 0;  // lanes past the end re-read element 0 and multiply it by zero columns
            gv[k] = ok ? gB[off[k]] : 0.f;
            sv[k] = ok ? sB[off[k]] : 0.f;
            yv[k] = ok ? yB[off[k]] : 0.f;
        }
    }
#pragma unroll 1
    for (int i = i_begin; i < i_end; ++i) {  // stored pairs, then the candidate pair
        const int p = i < len ? (head + i) % m1 : cand;
        const float* sp = S + (int64_t)p * n + blk;
        const float* yp = Y + (int64_t)p * n + blk;
        float ls[LB_EPT], ly[LB_EPT];
        if constexpr (VEC) {
#pragma unroll
            for (int k = 0; k < LB_EPT; k += 4) {  // (off[k] is a multiple of 4: rem % 4 == 0)
                const float4 a = LB_STREAM_LOAD(reinterpret_cast<const float4*>(sp + off[k]));
                const float4 b = LB_STREAM_LOAD(reinterpret_cast<const float4*>(yp + off[k]));
                ls[k] = a.x, ls[k + 1] = a.y, ls[k + 2] = a.z, ls[k + 3] = a.w;
                ly[k] = b.x, ly[k + 1] = b.y, ly[k + 2] = b.z, ly[k + 3] = b.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < LB_EPT; ++k) {
                ls[k] = LB_STREAM_LOAD(sp + off[k]);
                ly[k] = LB_STREAM_LOAD(yp + off[k]);
            }
        }
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, b0 = 0.f, b1 = 0.f, b2 = 0.f;
#pragma unroll
        for (int k = 0; k < LB_EPT; ++k) {
            a0 = fmaf(ls[k], sv[k], a0);
            a1 = fmaf(ls[k], yv[k], a1);
            a2 = fmaf(ls[k], gv[k], a2);
            b0 = fmaf(ly[k], sv[k], b0);
            b1 = fmaf(ly[k], yv[k], b1);
            b2 = fmaf(ly[k], gv[k], b2);
        }
        wave_reduce3_store(a0, a1, a2, mine + 4 * p, lane);
        wave_reduce3_store(b0, b1, b2, mine + 4 * (m1 + p), lane);
    }
    if (blockIdx.y == 0) {
        float sg = 0.f, yg = 0.f, gg = 0.f, g1 = 0.f, gm = 0.f;
#pragma unroll
        for (int k = 0; k < LB_EPT; ++k) {
            sg = fmaf(sv[k], gv[k], sg);
            yg = fmaf(yv[k], gv[k], yg);
            gg = fmaf(gv[k], gv[k], gg);
            g1 += fabsf(gv[k]);
            gm = fmaxf(gm, fabsf(gv[k]));
        }
        wave_reduce3_store(sg, yg, gg, mine + 4 * (2 * m1), lane);
        g1 = wave_sum(g1);
        gm = wave_max_nonneg(gm);
        if (lane == 0) {
            mine[4 * (2 * m1) + 3] = g1;
            atomicMax(&hdr->gmax_bits, __float_as_uint(gm));  // a maximum does not depend on the order: deterministic
        }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * nb_ids * 4;
    // this workgroup's entries: (pair, s | y, component) of its pairs, and the row of g for group 0
    for (int e = tid; e < (i_end - i_begin) * 8; e += 256) {
        const int i = i_begin + (e >> 3);
        const int p = i < len ? (head + i) % m1 : cand;
        const int o = (((e >> 2) & 1) * m1 + p) * 4 + (e & 3);
        out[o] = (lds[o] + lds[nb_ids * 4 + o]) + (lds[2 * nb_ids * 4 + o] + lds[3 * nb_ids * 4 + o]);
    }
    if (blockIdx.y == 0 && tid < 4) {
        const int o = 2 * m1 * 4 + tid;
        out[o] = (lds[o] + lds[nb_ids * 4 + o]) + (lds[2 * nb_ids * 4 + o] + lds[3 * nb_ids * 4 + o]);
    }
}

// One workgroup per id: dots[id][c] = sum over workgroups of partial[wg][id][c], fixed order, fp64.
__global__ void __launch_bounds__(256)
lbfgs_finish_dots_kernel(const LbfgsHeader* __restrict__ hdr, const float* __restrict__ partial, double* __restrict__ dots,
                         int nwg, int nb_ids, int m1) {
    __shared__ double scratch[16];
    const int id = blockIdx.x;
    if (id < 2 * m1) {  // slots outside the ring (and not the candidate) were not visited by the sweep: their dots are zero
        const int slot = id % m1;
        if ((slot - hdr->head + m1) % m1 >= hdr->len && slot != hdr->cand) {
            if (threadIdx.x < 4) dots[id * 4 + threadIdx.x] = 0.0;
            return;
        }
    }
    double a[4] = {0, 0, 0, 0};
    for (int w = threadIdx.x; w < nwg; w += blockDim.x) {
        const float* p = partial + ((size_t)w * nb_ids + id) * 4;
        a[0] += p[0];
        a[1] += p[1];
        a[2] += p[2];
        a[3] += p[3];
    }
    for (int c = 0; c < 4; ++c) {
        const double v = block_sum(a[c], scratch);
        if (threadIdx.x == 0) dots[id * 4 + c] = v;
    }
}

// Sum of a double over the wave with DPP row shifts / row broadcasts (no LDS round trips); the result is uniform.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add(double v) {
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const int lo2 = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi2, lo2);
}
__device__ inline double wave_total_f64(double v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1   (inclusive scan inside each row of 16 lanes)
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1, 3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2, 3 ; lane 63 holds the total
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}

// The recursion in coefficient space, one wave.  The 2*len steps form one dependent chain, so the step latency is what
// counts: delta stays in registers (lane l holds entries l + 64 q), the dot of a step is reduced with DPP, and the rows
// of M (which do not depend on delta) stream through a D-deep register ring so no step waits for global memory.
template <int NQ>
__global__ void __launch_bounds__(64)
lbfgs_coeffs_kernel(LbfgsHeader* __restrict__ hdr, const double* __restrict__ dots, double* __restrict__ M,
                    float* __restrict__ coef, int m1, int history, float lr, float tol_change, float tol_grad,
                    const float* __restrict__ loss) {
    extern __shared__ double sh[];  // alpha[m1 + 1], rinv[m1 + 1]  (slot m1: sink / zero for the padding steps)
    const int nb_ids = 2 * m1 + 1, gid = 2 * m1;
    double* alpha = sh;
    double* rinv = sh + m1 + 1;
    const int lane = threadIdx.x;
    int len = hdr->len, head = hdr->head, cand = hdr->cand;
    const bool first = hdr->n_iter == 0;
    double h_diag = hdr->h_diag;
    if (hdr->stopped) return;
    {
        // Stop tests of LBFGS.step that look at the new evaluation (lbfgs.py: `opt_cond` before the first direction and after
        // every re-evaluation, `d.mul(t).abs().max() <= tolerance_change`, `abs(loss - prev_loss) < tolerance_change`).
        // All of them end the step() call, i.e. x stays where the last move left it.
        const float gmax = __uint_as_float(hdr->gmax_bits), dmax = __uint_as_float(hdr->dmax_bits);
        const float cur = loss ? loss[0] : 0.f;
        bool stop = gmax <= tol_grad;
        if (!first) {
            stop = stop || dmax <= tol_change;
            if (loss) stop = stop || fabs((double)cur - (double)hdr->prev_loss) < (double)tol_change;
        }
        __syncthreads();  // every lane has read the header before lane 0 rewrites it
        if (lane == 0) {
            hdr->gmax_bits = 0u;
            hdr->dmax_bits = 0u;
            hdr->prev_loss = cur;
            if (stop) hdr->stopped = 1;
        }
        if (stop) return;
    }

    // 1. curvature test and ring update (lbfgs.py: `if ys > 1e-10`)
    const double ys = dots[(m1 + cand) * 4 + 0];  // y_new . s_new
    const double yy = dots[(m1 + cand) * 4 + 1];
    const bool commit = !first && (float)ys > 1e-10f;
    if (commit) {
        if (len == history) head = (head + 1) % m1;  // drop the oldest
        else ++len;
        h_diag = (double)((float)ys / (float)yy);
    }
    // 2. refresh M: columns of s_new / y_new (when committed) and of g
    for (int i = lane; i < nb_ids; i += 64) {
        const double ds = dots[i * 4 + 0], dy = dots[i * 4 + 1], dg = dots[i * 4 + 2];
        if (commit) {
            M[(size_t)i * nb_ids + cand] = ds;
            M[(size_t)cand * nb_ids + i] = ds;
            M[(size_t)i * nb_ids + m1 + cand] = dy;
            M[(size_t)(m1 + cand) * nb_ids + i] = dy;
        }
        M[(size_t)i * nb_ids + gid] = dg;
        M[(size_t)gid * nb_ids + i] = dg;
    }
    __syncthreads();
    __threadfence_block();
    for (int slot = lane; slot < m1; slot += 64) rinv[slot] = 1.0 / M[(size_t)(m1 + slot) * nb_ids + slot];  // ro = 1 / (y.s)
    if (lane == 0) rinv[m1] = 0.0;
    __syncthreads();

    // Unconditional loads (column index clamped; the matching delta entries are zero) so the ring is straight-line code
    // and the compiler can wait for exactly the oldest row instead of draining the queue.
    auto load_row = [&](int row, double (&r)[NQ]) {
        const double* mr = M + (size_t)row * nb_ids;
#pragma unroll
        for (int q = 0; q < NQ; ++q) r[q] = mr[min(lane + 64 * q, nb_ids - 1)];
    };
    double dl[NQ], grow[NQ];  // delta (q = -g to start with) and the row of g for the final g . d
#pragma unroll
    for (int q = 0; q < NQ; ++q) dl[q] = lane + 64 * q == gid ? -1.0 : 0.0;
    load_row(gid, grow);

    // step k < len: first loop, newest -> oldest (row of s_i); step k >= len: second loop, oldest -> newest (row of y_i)
    // Steps past `total` (padding up to a multiple of D) read the row of g and add zero.
    const int total = first ? 0 : 2 * len;
    auto row_of = [&](int k) { return k < len ? (head + len - 1 - k) % m1 : (k < total ? m1 + (head + k - len) % m1 : gid); };
    constexpr int D = 8;
    double ring[D][NQ];
#pragma unroll
    for (int u = 0; u < D; ++u) load_row(row_of(u), ring[u]);
    for (int k0 = 0; k0 < total; k0 += D) {
#pragma unroll
        for (int u = 0; u < D; ++u) {
            const int k = k0 + u;
            const bool live = k < total, second = k >= len;
            const int i = second ? k - len : len - 1 - k;
            const int pslot = live ? (head + i) % m1 : m1;
            const double ri = rinv[pslot];
            const double al_i = alpha[second && live ? i : m1];
            const double scale = k == len ? h_diag : 1.0;  // r = q * H_diag between the two loops
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                dl[q] *= scale;
                acc = fma(dl[q], ring[u][q], acc);
            }
            const double dot = wave_total_f64(acc) * ri;
            // first loop:  al[i] = old_stps[i].dot(q) * ro[i];  q.add_(old_dirs[i], alpha=-al[i])
            // second loop: be_i = old_dirs[i].dot(r) * ro[i];   r.add_(old_stps[i], alpha=al[i]-be_i)
            const int target = !live ? -1 : (second ? pslot : m1 + pslot);
            const double add = second ? al_i - dot : -dot;
            alpha[second || !live ? m1 : i] = dot;  // every lane stores the same value: no cross-lane hand-off to order
#pragma unroll
            for (int q = 0; q < NQ; ++q) dl[q] += lane + 64 * q == target ? add : 0.0;
            load_row(row_of(k + D), ring[u]);
        }
    }
    if (!first && len == 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q) dl[q] *= h_diag;
    }
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; ++q) acc = fma(dl[q], grow[q], acc);
    const double gtd = wave_total_f64(acc);  // g . d
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (lane + 64 * q < nb_ids) coef[lane + 64 * q] = (float)dl[q];
    if (lane == 0) {
        float t = lr;
        if (first) {
            const double g1 = dots[gid * 4 + 3];
            const float inv = (float)(1.0 / g1);
            t = (inv < 1.f ? inv : 1.f) * lr;  // min(1., 1./|g|_1) * lr
        }
        hdr->len = len;
        hdr->head = head;
        hdr->cand = (head + len) % m1;  // next free slot
        hdr->h_diag = h_diag;
        hdr->gtd = (float)gtd;
        hdr->t = t;
        if ((float)gtd > -tol_change) hdr->stopped = 1;  // `if gtd > -tolerance_change: break`
        hdr->n_iter = hdr->n_iter + 1;
    }
}

#ifdef LB_STAMP  // diagnostic build (tools/lbfgs_clock.py): shader-clock stamps at the phase boundaries of the coefficient kernel
__device__ unsigned long long g_lb_stamps[16];
#define LB_MARK(k)                                                      \
    do {                                                                \
        if (threadIdx.x == 0) g_lb_stamps[k] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define LB_MARK(k) do {} while (0)
#endif

__device__ __forceinline__ double lane_value_f64(double v, int l) {  // l uniform
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// The same recursion as two triangular substitutions (history <= 127).  In coefficient space a step of the first loop only
// changes ONE coefficient of q (that of y_i), so with a_j = ro_j (s_j . q) carried along for every pair j (lane j),
//     al_i = a_i ;  a_j -= al_i ro_j (s_j . y_i)  for the older pairs j                     (newest -> oldest)
// and likewise for the second loop with e_j = al_j - ro_j (y_j . r),
//     c_i = al_i - be_i = e_i ;  e_j -= c_i ro_j (s_i . y_j)  for the newer pairs j         (oldest -> newest)
// where y_j . r starts as H (y_j . q) = -H (y_j . g + sum_k al_k (y_j . y_k)), a parallel matrix-vector product.  A step is a
// lane broadcast and one FMA per lane against a column / row of the s.y block, which sits in LDS and is read four steps
// ahead: ~10 instructions per step instead of the dot-product-and-reduce chain of lbfgs_coeffs_kernel (a lone wave issues
// one instruction every ~5 cycles, so the instruction count IS the latency).  d = -H g - H sum al_k y_k + sum c_k s_k.
// All of M that the kernel needs is requested up front (M does not survive in L2 between iterations: a load costs ~2 us).
__global__ void __launch_bounds__(256)
lbfgs_coeffs_tri_kernel(LbfgsHeader* __restrict__ hdr, const double* __restrict__ dots, double* __restrict__ M,
                        float* __restrict__ coef, int m1, int history, float lr, float tol_change, float tol_grad,
                        const float* __restrict__ loss) {
    extern __shared__ double sh[];
    const int L = m1 | 1;  // odd row length: walking down a column is free of bank conflicts
    double* SY = sh;       // [m1][L], logical order (0 = oldest pair): SY[i][k] = s_i . y_k
    double* al = SY + (size_t)m1 * L;
    double* cc = al + m1;
    double* part = cc + m1;  // [384 + 16]
    const int nb_ids = 2 * m1 + 1, gid = 2 * m1;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int len = hdr->len, head = hdr->head;
    const int cand = hdr->cand;
    const bool first = hdr->n_iter == 0;
    double h_diag = hdr->h_diag;
    LB_MARK(0);
    if (hdr->stopped) return;
    {
        const float gmax = __uint_as_float(hdr->gmax_bits), dmax = __uint_as_float(hdr->dmax_bits);
        const float cur = loss ? loss[0] : 0.f;
        bool stop = gmax <= tol_grad;
        if (!first) {
            stop = stop || dmax <= tol_change;
            if (loss) stop = stop || fabs((double)cur - (double)hdr->prev_loss) < (double)tol_change;
        }
        __syncthreads();  // every thread has read the header before thread 0 rewrites it
        if (tid == 0) {
            hdr->gmax_bits = 0u;
            hdr->dmax_bits = 0u;
            hdr->prev_loss = cur;
            if (stop) hdr->stopped = 1;
        }
        if (stop) return;
    }
    const double ys = dots[(m1 + cand) * 4 + 0];
    const double yy = dots[(m1 + cand) * 4 + 1];
    const bool commit = !first && (float)ys > 1e-10f;
    if (commit) {
        if (len == history) head = (head + 1) % m1;
        else ++len;
        h_diag = (double)((float)ys / (float)yy);
    }
    LB_MARK(1);
    for (int i = tid; i < nb_ids; i += 256) {
        const double ds = dots[i * 4 + 0], dy = dots[i * 4 + 1], dg = dots[i * 4 + 2];
        if (commit) {
            M[(size_t)i * nb_ids + cand] = ds;
            M[(size_t)cand * nb_ids + i] = ds;
            M[(size_t)i * nb_ids + m1 + cand] = dy;
            M[(size_t)(m1 + cand) * nb_ids + i] = dy;
        }
        M[(size_t)i * nb_ids + gid] = dg;
        M[(size_t)gid * nb_ids + i] = dg;
    }
    __syncthreads();
    LB_MARK(2);
    const int total = first ? 0 : len;
    const int last = max(total - 1, 0);
    auto slot_of = [&](int i) {  // i < m1
        const int p = head + i;
        return p >= m1 ? p - m1 : p;
    };
    // this lane's pairs: j = lane and lane + 64 (clamped; lanes past `total` carry zeros)
    const int j0 = min(lane, last), j1 = min(lane + 64, last);
    const int p0 = slot_of(j0), p1 = slot_of(j1);
    const bool v0 = lane < total, v1 = lane + 64 < total;
    constexpr int RMAX = 32;  // rows of the s.y block per wave: ceil(127 / 4)
    {   // s.y block -> LDS: wave w takes the rows w, w + 4, ...; all requests first, then the writes
        double t0[RMAX], t1[RMAX];
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int i = min(wave + 4 * q, last);
            const double* row = M + (size_t)slot_of(i) * nb_ids + m1;
            t0[q] = row[p0];
            t1[q] = row[p1];
        }
#pragma unroll
        for (int q = 0; q < RMAX; ++q) {
            const int i = wave + 4 * q;
            if (i < total) {
                SY[(size_t)i * L + j0] = t0[q];
                if (v1) SY[(size_t)i * L + j1] = t1[q];
            }
        }
    }
    // (only wave 0 uses these; the loads are issued by every wave, which costs nothing)
    const double rj0 = v0 ? 1.0 / M[(size_t)(m1 + p0) * nb_ids + p0] : 0.0;  // ro_j = 1 / (y_j . s_j)
    const double rj1 = v1 ? 1.0 / M[(size_t)(m1 + p1) * nb_ids + p1] : 0.0;
    const double sg0 = M[(size_t)p0 * nb_ids + gid], sg1 = M[(size_t)p1 * nb_ids + gid];
    const double yg0 = M[(size_t)(m1 + p0) * nb_ids + gid], yg1 = M[(size_t)(m1 + p1) * nb_ids + gid];
    __syncthreads();
    LB_MARK(3);
    constexpr int D = 4;
    constexpr int KMAX = 43;  // ceil(127 / 3): pairs per wave in the y.y product
    double yyv[KMAX][2];      // waves 1-3: (y_k . y_j) for k = wave - 1 + 3 q, j = lane and lane + 64
    // LDS byte addresses of this lane's column / row walks
    const double* colp0 = SY + (size_t)j0 * L;
    const double* colp1 = SY + (size_t)j1 * L;
    if (wave == 0) {
        if (total > 0) {  // first loop: newest -> oldest
            double a0 = -rj0 * sg0, a1 = -rj1 * sg1;
            // steps whose pair sits in the upper register (i >= 64), then the lower one; each range in rounds of D with the
            // columns requested one round ahead, the last (partial) round one step at a time
            auto run = [&](int i_hi, int i_lo, bool upper) {  // i = i_hi ... i_lo, descending
                int i = i_hi;
                for (; i - i_lo + 1 >= D; i -= D) {
                    double c0[D], c1[D];
#pragma unroll
                    for (int u = 0; u < D; ++u) {
                        c0[u] = colp0[i - u];
                        c1[u] = colp1[i - u];
                    }
#pragma unroll
                    for (int u = 0; u < D; ++u) {
                        const double m0 = rj0 * c0[u], m1v = rj1 * c1[u];
                        const double v = lane_value_f64(upper ? a1 : a0, (i - u) & 63);
                        a0 = fma(-v, m0, a0);
                        a1 = fma(-v, m1v, a1);
                        al[i - u] = v;  // every lane stores the same value
                    }
                }
                for (; i >= i_lo; --i) {
                    const double m0 = rj0 * colp0[i], m1v = rj1 * colp1[i];
                    const double v = lane_value_f64(upper ? a1 : a0, i & 63);
                    a0 = fma(-v, m0, a0);
                    a1 = fma(-v, m1v, a1);
                    al[i] = v;
                }
            };
            if (total > 64) run(total - 1, 64, true);
            run(min(total, 64) - 1, 0, false);
        }
    } else {  // meanwhile the other waves pull the y.y block out of M (it is only needed once al is complete)
        const double* c0 = M + (size_t)m1 * nb_ids + m1 + p0;
        const double* c1 = M + (size_t)m1 * nb_ids + m1 + p1;
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
            const int k = min(wave - 1 + 3 * q, last);
            const size_t ro = (size_t)slot_of(k) * nb_ids;
            yyv[q][0] = c0[ro];
            yyv[q][1] = c1[ro];
        }
    }
    __syncthreads();
    LB_MARK(4);
    if (wave > 0) {  // sum_k al_k (y_j . y_k): each wave sums its third of the k range
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < KMAX; ++q) {
            const int k = wave - 1 + 3 * q;
            const double a = k < total ? al[k] : 0.0;
            s0 = fma(a, yyv[q][0], s0);
            s1 = fma(a, yyv[q][1], s1);
        }
        part[(wave - 1) * 128 + lane] = s0;
        part[(wave - 1) * 128 + lane + 64] = s1;
    }
    __syncthreads();
    LB_MARK(5);
    if (wave == 0 && total > 0) {  // second loop: oldest -> newest
        // y_j . r starts as H (y_j . q) = -H (y_j . g + sum_k al_k (y_j . y_k))
        const double b0 = -h_diag * (yg0 + ((part[lane] + part[128 + lane]) + part[256 + lane]));
        const double b1 = -h_diag * (yg1 + ((part[lane + 64] + part[128 + lane + 64]) + part[256 + lane + 64]));
        double e0 = v0 ? al[j0] - rj0 * b0 : 0.0, e1 = v1 ? al[j1] - rj1 * b1 : 0.0;
        const double* rowp0 = SY + j0;
        const double* rowp1 = SY + j1;
        auto run = [&](int i_lo, int i_hi, bool upper) {  // i = i_lo ... i_hi, ascending
            int i = i_lo;
            for (; i_hi - i + 1 >= D; i += D) {
                double c0[D], c1[D];
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    c0[u] = rowp0[(size_t)(i + u) * L];
                    c1[u] = rowp1[(size_t)(i + u) * L];
                }
#pragma unroll
                for (int u = 0; u < D; ++u) {
                    const double m0 = rj0 * c0[u], m1v = rj1 * c1[u];
                    const double c = lane_value_f64(upper ? e1 : e0, (i + u) & 63);
                    e0 = fma(-c, m0, e0);
                    e1 = fma(-c, m1v, e1);
                    cc[i + u] = c;
                }
            }
            for (; i <= i_hi; ++i) {
                const double m0 = rj0 * rowp0[(size_t)i * L], m1v = rj1 * rowp1[(size_t)i * L];
                const double c = lane_value_f64(upper ? e1 : e0, i & 63);
                e0 = fma(-c, m0, e0);
                e1 = fma(-c, m1v, e1);
                cc[i] = c;
            }
        };
        run(0, min(total, 64) - 1, false);
        if (total > 64) run(64, total - 1, true);
    }
    __syncthreads();
    LB_MARK(6);
    // coefficients of d over the basis (s slots, y slots, g) and g . d
    const double hq = first ? 1.0 : h_diag;
    double gtd_part = 0.0;
    for (int id = tid; id < nb_ids; id += 256) {
        double v = 0.0;
        if (id == gid) {
            v = -hq;
        } else {
            int i = (id < m1 ? id : id - m1) - head;
            i = i < 0 ? i + m1 : i;
            if (i < total) v = id < m1 ? cc[i] : -h_diag * al[i];
        }
        if (v != 0.0) gtd_part += v * M[(size_t)id * nb_ids + gid];
        coef[id] = (float)v;
    }
    const double gtd = block_sum(gtd_part, part + 384);
    if (tid == 0) {
        float t = lr;
        if (first) {
            const double g1 = dots[gid * 4 + 3];
            const float inv = (float)(1.0 / g1);
            t = (inv < 1.f ? inv : 1.f) * lr;
        }
        hdr->len = len;
        hdr->head = head;
        hdr->cand = (head + len) % m1;
        hdr->h_diag = h_diag;
        hdr->gtd = (float)gtd;
        hdr->t = t;
        if ((float)gtd > -tol_change) hdr->stopped = 1;
        hdr->n_iter = hdr->n_iter + 1;
    }
    LB_MARK(7);
}

// Sweep 2: d = sum_id coef[id] * b_id ; x += t * d unless stopped.
// EPT elements per thread: 16 for large vectors; fewer for small ones so that the launch still covers the chip (the per-element
// sums do not depend on it).
template <int EPT>
__global__ void __launch_bounds__(256)
lbfgs_combine_kernel(const LbfgsHeader* __restrict__ hdr, const float* __restrict__ coef, const float* __restrict__ g,
                     const float* __restrict__ S, const float* __restrict__ Y, float* __restrict__ d, float* __restrict__ x,
                     int64_t n, int m1, unsigned* __restrict__ dmax_bits) {
    const int len = hdr->len, head = hdr->head, stopped = hdr->stopped;
    const float t = hdr->t;
    const int64_t blk = (int64_t)blockIdx.x * (256 * EPT);
    const int rem = (int)min((int64_t)(256 * EPT), n - blk);
    const int tid = threadIdx.x;
    float acc[EPT];
    const float cg = coef[2 * m1];
    const float* gB = g + blk;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + 256 * k;
        acc[k] = e < rem ? cg * gB[e] : 0.f;
    }
#pragma unroll EPT >= 16 ? 2 : 8
    for (int i = 0; i < len; ++i) {
        const int p = (head + i) % m1;
        const float cs = coef[p], cy = coef[m1 + p];
        const float* sp = S + (int64_t)p * n + blk;
        const float* yp = Y + (int64_t)p * n + blk;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int e = tid + 256 * k;
            const int ec = e < rem ? e : 0;
            acc[k] = fmaf(cy, LB_STREAM_LOAD(yp + ec), fmaf(cs, LB_STREAM_LOAD(sp + ec), acc[k]));  // lanes past the end accumulate garbage, never stored
        }
    }
    float* dB = d + blk;
    float* xB = x + blk;
    float dm = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int e = tid + 256 * k;
        if (e < rem) {
            dB[e] = acc[k];
            dm = fmaxf(dm, fabsf(acc[k] * t));             // d.mul(t).abs().max()
            if (!stopped) xB[e] = fmaf(t, acc[k], xB[e]);  // p.add_(d, alpha=t)
        }
    }
    dm = wave_max_nonneg(dm);
    if ((tid & 63) == 0 && !stopped) atomicMax(dmax_bits, __float_as_uint(dm));
}

// Sweep 2 for small vectors (count % 4 == 0): one float4 per thread, 64-thread workgroups - wide loads AND enough waves to
// cover the chip; per element the same sum in the same order as above.
__global__ void __launch_bounds__(256)
lbfgs_combine_v4_kernel(const LbfgsHeader* __restrict__ hdr, const float* __restrict__ coef, const float* __restrict__ g,
                        const float* __restrict__ S, const float* __restrict__ Y, float* __restrict__ d, float* __restrict__ x,
                        int64_t n, int m1, unsigned* __restrict__ dmax_bits) {
    const int len = hdr->len, head = hdr->head, stopped = hdr->stopped;
    const float t = hdr->t;
    const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const bool ok = e < n;
    const int64_t ec = ok ? e : 0;
    const float cg = coef[2 * m1];
    const float4 gv = *reinterpret_cast<const float4*>(g + ec);
    float4 acc = make_float4(cg * gv.x, cg * gv.y, cg * gv.z, cg * gv.w);
#pragma unroll 8
    for (int i = 0; i < len; ++i) {
        const int p = (head + i) % m1;
        const float cs = coef[p], cy = coef[m1 + p];
        const float4 sv = LB_STREAM_LOAD(reinterpret_cast<const float4*>(S + (int64_t)p * n + ec));
        const float4 yv = LB_STREAM_LOAD(reinterpret_cast<const float4*>(Y + (int64_t)p * n + ec));
        acc.x = fmaf(cy, yv.x, fmaf(cs, sv.x, acc.x));
        acc.y = fmaf(cy, yv.y, fmaf(cs, sv.y, acc.y));
        acc.z = fmaf(cy, yv.z, fmaf(cs, sv.z, acc.z));
        acc.w = fmaf(cy, yv.w, fmaf(cs, sv.w, acc.w));
    }
    float dm = 0.f;
    if (ok) {
        *reinterpret_cast<float4*>(d + e) = acc;
        dm = fmaxf(fmaxf(fabsf(acc.x * t), fabsf(acc.y * t)), fmaxf(fabsf(acc.z * t), fabsf(acc.w * t)));
        if (!stopped) {
            float4 xv = *reinterpret_cast<const float4*>(x + e);
            xv.x = fmaf(t, acc.x, xv.x);
            xv.y = fmaf(t, acc.y, xv.y);
            xv.z = fmaf(t, acc.z, xv.z);
            xv.w = fmaf(t, acc.w, xv.w);
            *reinterpret_cast<float4*>(x + e) = xv;
        }
    }
    dm = wave_max_nonneg(dm);
    if ((threadIdx.x & 63) == 0 && !stopped) atomicMax(dmax_bits, __float_as_uint(dm));
}

__global__ void lbfgs_status_kernel(const LbfgsHeader* __restrict__ hdr, float* __restrict__ out) {
    if (threadIdx.x == 0) {
        out[0] = (float)hdr->n_iter;
        out[1] = (float)hdr->len;
        out[2] = (float)hdr->stopped;
        out[3] = hdr->gtd;
        out[4] = hdr->t;
    }
}

}  // namespace maua

using namespace maua;

extern "C" {

size_t maua_lbfgs_state_bytes(int64_t count, int history) {
    if (count <= 0 || history <= 0 || history > 254 || count >= (1ll << 36)) return 0;
    return lbfgs_layout(count, history).total;
}

int maua_lbfgs_init(void* state, size_t state_bytes, int64_t count, int history, maua_stream_t stream) {
    MAUA_REQUIRE(state && count > 0 && count < (1ll << 36) && history > 0 && history <= 254, MAUA_E_INVAL, "lbfgs_init: bad args");
    MAUA_REQUIRE(2 * (history + 1) + 1 <= 512, MAUA_E_UNSUPPORTED, "lbfgs_init: history %d too large (at most 254)", history);
    const LbfgsLayout L = lbfgs_layout(count, history);
    MAUA_REQUIRE(state_bytes >= L.total, MAUA_E_WORKSPACE, "lbfgs_init: state %zu < %zu bytes", state_bytes, L.total);
    char* b = (char*)state;
    hipLaunchKernelGGL(lbfgs_init_kernel, dim3(64), dim3(256), 0, (hipStream_t)stream, (LbfgsHeader*)b,
                       (double*)(b + L.off_M), L.nb_ids, (float*)(b + L.off_coef));
    return check_launch("lbfgs_init_kernel");
}

int maua_lbfgs_iterate(void* state, float* x, const float* grad, const float* loss, int64_t count, int history, float lr,
                       float tolerance_change, float tolerance_grad, maua_stream_t stream) {
    MAUA_REQUIRE(state && x && grad && count > 0 && count < (1ll << 36) && history > 0 && history <= 254, MAUA_E_INVAL, "lbfgs_iterate: bad args");
    const LbfgsLayout L = lbfgs_layout(count, history);
    char* b = (char*)state;
    LbfgsHeader* hdr = (LbfgsHeader*)b;
    float* coef = (float*)(b + L.off_coef);
    double* dots = (double*)(b + L.off_dots);
    double* M = (double*)(b + L.off_M);
    float* partial = (float*)(b + L.off_partial);
    float* g_prev = (float*)(b + L.off_gprev);
    float* d = (float*)(b + L.off_d);
    float* S = (float*)(b + L.off_S);
    float* Y = (float*)(b + L.off_Y);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds1 = sizeof(float) * 4 * L.nb_ids * 4;
    // lbfgs_coeffs_kernel<8> keeps the 2*history+3 coefficients in 8 registers x 64 lanes
    MAUA_REQUIRE(lds1 <= 64 * 1024 && L.nb_ids <= 512, MAUA_E_UNSUPPORTED, "lbfgs_iterate: history %d too large (at most 254)", history);
    int pg = (int)((count + 1023) / 1024);
    if (pg > 2048) pg = 2048;
    hipLaunchKernelGGL(lbfgs_pair_kernel, dim3(pg), dim3(256), 0, s, hdr, grad, g_prev, d, S, Y, count);
    int rc = check_launch("lbfgs_pair_kernel");
    if (rc) return rc;
    int groups = (1024 + L.nwg - 1) / L.nwg;  // pair-loop groups: aim at ~4 workgroups per CU
    if (groups > L.m1) groups = L.m1;
    const bool aligned = count % 4 == 0 && (((uintptr_t)x | (uintptr_t)grad) & 15) == 0;
    const bool vec_on = tuning("lbfgs_vec", 1) != 0;
    if (aligned && vec_on)
        hipLaunchKernelGGL(lbfgs_pair_dots_kernel<true>, dim3(L.nwg, groups), dim3(256), lds1, s, hdr, grad, S, Y, partial, count, L.m1);
    else
        hipLaunchKernelGGL(lbfgs_pair_dots_kernel<false>, dim3(L.nwg, groups), dim3(256), lds1, s, hdr, grad, S, Y, partial, count, L.m1);
    rc = check_launch("lbfgs_pair_dots_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(lbfgs_finish_dots_kernel, dim3(L.nb_ids), dim3(256), 0, s, hdr, partial, dots, L.nwg, L.nb_ids, L.m1);
    rc = check_launch("lbfgs_finish_dots_kernel");
    if (rc) return rc;
    const size_t lds3 = sizeof(double) * 2 * (L.m1 + 1);
    const size_t lds_tri = sizeof(double) * ((size_t)L.m1 * (L.m1 | 1) + 2 * (size_t)L.m1 + 384 + 16);
    const bool tri_on = tuning("lbfgs_tri", 1) != 0;
    if (tri_on && L.m1 <= 128 && lds_tri <= 144 * 1024) {
        static unsigned long long attr_set = 0;
        (void)opt_in_dynamic_lds(reinterpret_cast<const void*>(lbfgs_coeffs_tri_kernel), 144 * 1024, &attr_set);
        hipLaunchKernelGGL(lbfgs_coeffs_tri_kernel, dim3(1), dim3(256), lds_tri, s, hdr, dots, M, coef, L.m1, history, lr,
                           tolerance_change, tolerance_grad, loss);
    } else if (L.nb_ids <= 256)
        hipLaunchKernelGGL(lbfgs_coeffs_kernel<4>, dim3(1), dim3(64), lds3, s, hdr, dots, M, coef, L.m1, history, lr,
                           tolerance_change, tolerance_grad, loss);
    else
        hipLaunchKernelGGL(lbfgs_coeffs_kernel<8>, dim3(1), dim3(64), lds3, s, hdr, dots, M, coef, L.m1, history, lr,
                           tolerance_change, tolerance_grad, loss);
    rc = check_launch("lbfgs_coeffs_kernel");
    if (rc) return rc;
    auto combine = [&](auto kernel, int ept) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)((count + 256 * ept - 1) / (256 * ept))), dim3(256), 0, s, hdr, coef, grad, S, Y, d,
                           x, count, L.m1, &hdr->dmax_bits);
    };
    if (count >= 256 * 16 * 512) combine(lbfgs_combine_kernel<16>, 16);  // (large vectors: both forms run at the HBM rate)
    else if (aligned && vec_on)      // small ones: one float4 per thread, 64-thread workgroups - wide loads and enough waves
        hipLaunchKernelGGL(lbfgs_combine_v4_kernel, dim3((unsigned)((count / 4 + 63) / 64)), dim3(64), 0, s, hdr, coef, grad, S, Y, d,
                           x, count, L.m1, &hdr->dmax_bits);
    else if (count >= 256 * 4 * 512) combine(lbfgs_combine_kernel<4>, 4);
    else combine(lbfgs_combine_kernel<1>, 1);
    return check_launch("lbfgs_combine_kernel");
}

#ifdef LB_STAMP
int maua_lbfgs_read_stamps(unsigned long long* host16) {
    return (int)hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_lb_stamps), sizeof(unsigned long long) * 16);
}
#endif

int maua_lbfgs_status(const void* state, int64_t count, int history, float* out5, maua_stream_t stream) {
    MAUA_REQUIRE(state && out5 && count > 0 && history > 0 && history <= 254, MAUA_E_INVAL, "lbfgs_status: bad args");
    hipLaunchKernelGGL(lbfgs_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const LbfgsHeader*)state, out5);
    return check_launch("lbfgs_status_kernel");
}

}  // extern "C"
