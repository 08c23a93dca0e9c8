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
#include "common.hpp"

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
__global__ void __launch_bounds__(256)
lbfgs_pair_dots_kernel(LbfgsHeader* __restrict__ hdr, const float* __restrict__ g, const float* __restrict__ S,
                       const float* __restrict__ Y, float* __restrict__ partial, int64_t n, int m1) {
    extern __shared__ float lds[];  // [4 waves][nb_ids][4]
    const int nb_ids = 2 * m1 + 1;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4 * nb_ids * 4; i += 256) lds[i] = 0.f;
    __syncthreads();
    float* mine = lds + (size_t)wave * nb_ids * 4;

    const int cand = hdr->cand, len = hdr->len, head = hdr->head;
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
            const int e = tid + 256 * k;
            const bool ok = e < rem;
            off[k] = ok ? e : 0;  // lanes past the end re-read element 0 and multiply it by zero columns
            gv[k] = ok ? gB[off[k]] : 0.f;
            sv[k] = ok ? sB[off[k]] : 0.f;
            yv[k] = ok ? yB[off[k]] : 0.f;
        }
    }
#pragma unroll 1
    for (int i = 0; i <= len; ++i) {  // stored pairs, then the candidate pair
        const int p = i < len ? (head + i) % m1 : cand;
        const float* sp = S + (int64_t)p * n + blk;
        const float* yp = Y + (int64_t)p * n + blk;
        float ls[LB_EPT], ly[LB_EPT];
#pragma unroll
        for (int k = 0; k < LB_EPT; ++k) {
            ls[k] = sp[off[k]];
            ly[k] = yp[off[k]];
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
    {
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
    for (int i = tid; i < nb_ids * 4; i += 256)
        out[i] = (lds[i] + lds[nb_ids * 4 + i]) + (lds[2 * nb_ids * 4 + i] + lds[3 * nb_ids * 4 + i]);
}

// One workgroup per id: dots[id][c] = sum over workgroups of partial[wg][id][c], fixed order, fp64.
__global__ void __launch_bounds__(256)
lbfgs_finish_dots_kernel(const float* __restrict__ partial, double* __restrict__ dots, int nwg, int nb_ids) {
    __shared__ double scratch[16];
    const int id = blockIdx.x;
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

// Sweep 2: d = sum_id coef[id] * b_id ; x += t * d unless stopped.
__global__ void __launch_bounds__(256)
lbfgs_combine_kernel(const LbfgsHeader* __restrict__ hdr, const float* __restrict__ coef, const float* __restrict__ g,
                     const float* __restrict__ S, const float* __restrict__ Y, float* __restrict__ d, float* __restrict__ x,
                     int64_t n, int m1, unsigned* __restrict__ dmax_bits) {
    const int len = hdr->len, head = hdr->head, stopped = hdr->stopped;
    const float t = hdr->t;
    const int64_t blk = (int64_t)blockIdx.x * LB_WG;
    const int rem = (int)min((int64_t)LB_WG, n - blk);
    const int tid = threadIdx.x;
    float acc[LB_EPT];
    const float cg = coef[2 * m1];
    const float* gB = g + blk;
#pragma unroll
    for (int k = 0; k < LB_EPT; ++k) {
        const int e = tid + 256 * k;
        acc[k] = e < rem ? cg * gB[e] : 0.f;
    }
#pragma unroll 2
    for (int i = 0; i < len; ++i) {
        const int p = (head + i) % m1;
        const float cs = coef[p], cy = coef[m1 + p];
        const float* sp = S + (int64_t)p * n + blk;
        const float* yp = Y + (int64_t)p * n + blk;
#pragma unroll
        for (int k = 0; k < LB_EPT; ++k) {
            const int e = tid + 256 * k;
            const int ec = e < rem ? e : 0;
            acc[k] = fmaf(cy, yp[ec], fmaf(cs, sp[ec], acc[k]));  // lanes past the end accumulate garbage, never stored
        }
    }
    float* dB = d + blk;
    float* xB = x + blk;
    float dm = 0.f;
#pragma unroll
    for (int k = 0; k < LB_EPT; ++k) {
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
    hipLaunchKernelGGL(lbfgs_pair_dots_kernel, dim3(L.nwg), dim3(256), lds1, s, hdr, grad, S, Y, partial, count, L.m1);
    rc = check_launch("lbfgs_pair_dots_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(lbfgs_finish_dots_kernel, dim3(L.nb_ids), dim3(256), 0, s, partial, dots, L.nwg, L.nb_ids);
    rc = check_launch("lbfgs_finish_dots_kernel");
    if (rc) return rc;
    const size_t lds3 = sizeof(double) * 2 * (L.m1 + 1);
    if (L.nb_ids <= 256)
        hipLaunchKernelGGL(lbfgs_coeffs_kernel<4>, dim3(1), dim3(64), lds3, s, hdr, dots, M, coef, L.m1, history, lr,
                           tolerance_change, tolerance_grad, loss);
    else
        hipLaunchKernelGGL(lbfgs_coeffs_kernel<8>, dim3(1), dim3(64), lds3, s, hdr, dots, M, coef, L.m1, history, lr,
                           tolerance_change, tolerance_grad, loss);
    rc = check_launch("lbfgs_coeffs_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(lbfgs_combine_kernel, dim3(L.nwg), dim3(256), 0, s, hdr, coef, grad, S, Y, d, x, count, L.m1,
                       &hdr->dmax_bits);
    return check_launch("lbfgs_combine_kernel");
}

int maua_lbfgs_status(const void* state, int64_t count, int history, float* out5, maua_stream_t stream) {
    MAUA_REQUIRE(state && out5 && count > 0 && history > 0 && history <= 254, MAUA_E_INVAL, "lbfgs_status: bad args");
    hipLaunchKernelGGL(lbfgs_status_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const LbfgsHeader*)state, out5);
    return check_launch("lbfgs_status_kernel");
}

}  // extern "C"
