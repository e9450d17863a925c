// HBM-bound kernels of the LLaVA hot path for gfx950: norms, RoPE, SwiGLU, GELU, cross-entropy, embedding
// splice, CLIP embeddings, transposes, reductions, AdamW.  All loads/stores are 16-byte vectors (8 bf16 per lane);
// reductions are wave shuffles + one LDS hop; no atomics (results are run-to-run deterministic).
// Reference call sites are listed per entry point in include/radvlm_hip.h.
#include "common.h"
#include "radvlm_hip.h"

namespace {

constexpr int TPB = 256;

DEVINL void ld8(const bf16* p, float (&v)[8]) {
    const bf16x8 t = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = bf2f(t[i]);
}
DEVINL void st8(bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = f2bf(v[i]);
    *(bf16x8*)p = t;
}
DEVINL float rbf(float x) { return bf2f(f2bf(x)); }  // round through bf16

// ------------------------------------------------------------------------------------------------ RMSNorm
// one 256-thread block per row; d <= 8192, d % 8 == 0
__global__ __launch_bounds__(TPB) void rmsnorm_fwd_kernel(const bf16* x, const bf16* w, bf16* y, float* rstd_out, int d, float eps) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    const bf16* xr = x + row * d;
    float xv[4][8];
    float ss = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            ld8(xr + e, xv[p]);
#pragma unroll
            for (int i = 0; i < 8; ++i) ss += xv[p][i] * xv[p][i];
        }
    }
    ss = block_sum<4>(ss, red);
    const float rstd = rsqrtf(ss / (float)d + eps);
    if (threadIdx.x == 0 && rstd_out) rstd_out[row] = rstd;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            float wv[8], o[8];
            ld8(w + e, wv);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = wv[i] * rbf(xv[p][i] * rstd);
            st8(y + row * d + e, o);
        }
    }
}

__global__ __launch_bounds__(TPB) void rmsnorm_bwd_kernel(const bf16* dy, const bf16* x, const bf16* w, const float* rstd,
                                                          bf16* dx, int dx_add, float* dw_partial, int rows, int d) {
    // One row per block iteration; the NEXT row's three streams (x, dy and, when accumulating, dx) are fetched before this row's
    // block-wide reduction, so two rows of loads are in flight per block (the reduction + its barriers used to sit between a row's
    // loads and the next row's: 3.2 TB/s).
    __shared__ float red[4];
    float dw[4][8];
    float wv[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dw[p][i] = 0.f; wv[p][i] = 0.f; }
        if (e < d) ld8(w + e, wv[p]);
    }
    const bf16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 nx[4], ny[4], nd[4];
    auto fetch = [&](long row) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = (p * TPB + threadIdx.x) * 8;
            const bool ok = e < d && row < rows;
            nx[p] = ok ? *(const bf16x8*)(x + row * d + e) : z8;
            ny[p] = ok ? *(const bf16x8*)(dy + row * d + e) : z8;
            nd[p] = (ok && dx_add) ? *(const bf16x8*)(dx + row * d + e) : z8;
        }
    };
    fetch(blockIdx.x);
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float rs = rstd[row];
        float xh[4][8], g[4][8], o[4][8];
        float dot = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float dv = bf2f(ny[p][i]);
                xh[p][i] = bf2f(nx[p][i]) * rs;
                g[p][i] = dv * wv[p][i];
                dot += g[p][i] * xh[p][i];
                dw[p][i] += dv * xh[p][i];
                o[p][i] = bf2f(nd[p][i]);
            }
        fetch(row + gridDim.x);
        dot = block_sum<4>(dot, red) / (float)d;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = (p * TPB + threadIdx.x) * 8;
            if (e < d) {
#pragma unroll
                for (int i = 0; i < 8; ++i) o[p][i] += rs * (g[p][i] - xh[p][i] * dot);
                st8(dx + row * d + e, o[p]);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            float* o = dw_partial + (long)blockIdx.x * d + e;
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = dw[p][i];
        }
    }
}

__global__ __launch_bounds__(TPB) void layernorm_fwd_kernel(const bf16* x, const bf16* w, const bf16* b, bf16* y, float* stats, int d, float eps) {
    __shared__ float red[4];
    const long row = blockIdx.x;
    float xv[4][8];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            ld8(x + row * d + e, xv[p]);
#pragma unroll
            for (int i = 0; i < 8; ++i) s += xv[p][i];
        }
    }
    const float mean = block_sum<4>(s, red) / (float)d;
    float v = 0.f;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float t = xv[p][i] - mean; v += t * t; }
        }
    }
    const float rstd = rsqrtf(block_sum<4>(v, red) / (float)d + eps);
    if (stats && threadIdx.x == 0) { stats[2 * row] = mean; stats[2 * row + 1] = rstd; }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            float wv[8], bv[8], o[8];
            ld8(w + e, wv);
            ld8(b + e, bv);
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = (xv[p][i] - mean) * rstd * wv[i] + bv[i];
            st8(y + row * d + e, o);
        }
    }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)), g = dy * w;  partial[blk] = [sum dy*xhat (d) | sum dy (d)]
__global__ __launch_bounds__(TPB) void layernorm_bwd_kernel(const bf16* dy, const bf16* x, const bf16* w, const float* stats,
                                                            bf16* dx, int dx_add, float* partial, int rows, int d) {
    __shared__ float red[4];
    float dw[4][8], db[4][8], wv[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) { dw[p][i] = 0.f; db[p][i] = 0.f; }
        if (e < d) ld8(w + e, wv[p]);
    }
    for (long row = blockIdx.x; row < rows; row += gridDim.x) {
        const float mean = stats[2 * row], rs = stats[2 * row + 1];
        float xh[4][8], g[4][8];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = (p * TPB + threadIdx.x) * 8;
            if (e < d) {
                float xv[8], dv[8];
                ld8(x + row * d + e, xv);
                ld8(dy + row * d + e, dv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    xh[p][i] = (xv[i] - mean) * rs;
                    g[p][i] = dv[i] * wv[p][i];
                    s1 += g[p][i];
                    s2 += g[p][i] * xh[p][i];
                    dw[p][i] += dv[i] * xh[p][i];
                    db[p][i] += dv[i];
                }
            }
        }
        s1 = block_sum<4>(s1, red) / (float)d;
        s2 = block_sum<4>(s2, red) / (float)d;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int e = (p * TPB + threadIdx.x) * 8;
            if (e < d) {
                float o[8];
                if (dx_add) ld8(dx + row * d + e, o);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float v = rs * (g[p][i] - s1 - xh[p][i] * s2);
                    o[i] = dx_add ? o[i] + v : v;
                }
                st8(dx + row * d + e, o);
            }
        }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int e = (p * TPB + threadIdx.x) * 8;
        if (e < d) {
            float* o = partial + (long)blockIdx.x * 2 * d + e;
#pragma unroll
            for (int i = 0; i < 8; ++i) { o[i] = dw[p][i]; o[d + i] = db[p][i]; }
        }
    }
}

__global__ void quick_gelu_fwd_kernel(const bf16* x, bf16* y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], o[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = v[j] / (1.f + __expf(-1.702f * v[j]));
    st8(y + i * 8, o);
}
__global__ void quick_gelu_bwd_kernel(const bf16* dy, const bf16* x, bf16* dx, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], g[8], o[8];
    ld8(x + i * 8, v);
    ld8(dy + i * 8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float sg = 1.f / (1.f + __expf(-1.702f * v[j]));
        o[j] = g[j] * sg * (1.f + 1.702f * v[j] * (1.f - sg));
    }
    st8(dx + i * 8, o);
}

// ------------------------------------------------------------------------------------------------ column sums
// 32 columns x 32 row-lanes per 1024-thread block: each lane strides the rows by 32, then one LDS hop combines them (with 8 row-lanes
// and 1024 partial rows a call took 43 us of mostly load latency: 65 calls per step)
__global__ __launch_bounds__(1024) void colsum_f32_kernel(const float* in, int rows, int cols, bf16* out, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float s = 0.f;
    if (c < cols)
        for (int r = rl; r < rows; r += 32) s += in[(long)r * cols + c];
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < cols) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) t += red[i][cl];
        if (accumulate) t += bf2f(out[c]);
        out[c] = f2bf(t);
    }
}

__global__ __launch_bounds__(TPB) void colsum_partial_kernel(const bf16* x, long ld, int rows, int cols, float* partial) {
    const int c0 = (blockIdx.x * TPB + threadIdx.x) * 8;
    if (c0 >= cols) return;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
        float v[8];
        ld8(x + (long)r * ld + c0, v);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += v[i];
    }
    float* o = partial + (long)blockIdx.y * cols + c0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = acc[i];
}

// ------------------------------------------------------------------------------------------------ RoPE
__global__ void rope_kernel(bf16* x, long ld, const float* cs, const int* positions, int rows, int S, int heads, int hd, int nsec, int dir) {
    const int per_head = hd / 16;  // threads per head (8 pairs each)
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long total = (long)rows * nsec * heads * per_head;
    if (tid >= total) return;
    const int t = tid % per_head;
    const int h = (tid / per_head) % heads;
    const int sec = (tid / ((long)per_head * heads)) % nsec;
    const long row = tid / ((long)per_head * heads * nsec);
    const int pos = positions ? positions[row] : (int)(row % S);
    bf16* p = x + row * ld + (long)sec * heads * hd + h * hd + t * 8;
    float a[8], b[8];
    ld8(p, a);
    ld8(p + hd / 2, b);
    const float* c = cs + ((long)pos * (hd / 2) + t * 8) * 2;
    float oa[8], ob[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float co = c[2 * i], si = dir > 0 ? c[2 * i + 1] : -c[2 * i + 1];
        oa[i] = a[i] * co - b[i] * si;
        ob[i] = b[i] * co + a[i] * si;
    }
    st8(p, oa);
    st8(p + hd / 2, ob);
}

// ------------------------------------------------------------------------------------------------ SwiGLU / GELU
__global__ void swiglu_fwd_kernel(const bf16* gu, long ld_gu, bf16* act, long ld_act, int rows, int F) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = F / 8;
    if (tid >= (long)rows * per_row) return;
    const long r = tid / per_row;
    const int f = (tid % per_row) * 8;
    float g[8], u[8], o[8];
    ld8(gu + r * ld_gu + f, g);
    ld8(gu + r * ld_gu + F + f, u);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = rbf(g[i] / (1.f + __expf(-g[i]))) * u[i];
    st8(act + r * ld_act + f, o);
}

__global__ void swiglu_bwd_kernel(const bf16* dact, long ld_dact, const bf16* gu, long ld_gu, bf16* dgu, long ld_dgu, int rows, int F) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = F / 8;
    if (tid >= (long)rows * per_row) return;
    const long r = tid / per_row;
    const int f = (tid % per_row) * 8;
    float g[8], u[8], da[8], dg[8], du[8];
    ld8(gu + r * ld_gu + f, g);
    ld8(gu + r * ld_gu + F + f, u);
    ld8(dact + r * ld_dact + f, da);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(-g[i]));   // the same expression as the fused dgrad epilogue (gemm_bf16.hip): bit-identical results
        const float silu = g[i] * sg;
        du[i] = da[i] * silu;
        dg[i] = da[i] * u[i] * (sg * (1.f + g[i] * (1.f - sg)));
    }
    st8(dgu + r * ld_dgu + f, dg);
    st8(dgu + r * ld_dgu + F + f, du);
}

__global__ void dropout_kernel(const bf16* x, bf16* y, long n8, unsigned thr16, float scale, unsigned long long seed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], o[8];
    ld8(x + i * 8, v);
    const unsigned keep = rv_keep8(seed, (unsigned long long)i * 8, thr16);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (keep >> j) & 1 ? v[j] * scale : 0.f;
    st8(y + i * 8, o);
}

__global__ void dropout_add_kernel(const bf16* x, bf16* y, long n8, unsigned thr16, float scale, unsigned long long seed) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], o[8];
    ld8(x + i * 8, v);
    ld8(y + i * 8, o);
    const unsigned keep = rv_keep8(seed, (unsigned long long)i * 8, thr16);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] += (keep >> j) & 1 ? v[j] * scale : 0.f;
    st8(y + i * 8, o);
}

__global__ void gelu_fwd_kernel(const bf16* x, bf16* y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], o[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752f));
    st8(y + i * 8, o);
}
__global__ void gelu_bwd_kernel(const bf16* dy, const bf16* x, bf16* dx, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], g[8], o[8];
    ld8(x + i * 8, v);
    ld8(dy + i * 8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float cdf = 0.5f * (1.f + erff(v[j] * 0.70710678118654752f));
        const float pdf = 0.3989422804014327f * __expf(-0.5f * v[j] * v[j]);
        o[j] = g[j] * (cdf + v[j] * pdf);
    }
    st8(dx + i * 8, o);
}

__global__ void gelu_tanh_fwd_kernel(const bf16* x, bf16* y, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], o[8];
    ld8(x + i * 8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.5f * v[j] * (1.f + tanhf(0.7978845608028654f * (v[j] + 0.044715f * v[j] * v[j] * v[j])));
    st8(y + i * 8, o);
}
__global__ void gelu_tanh_bwd_kernel(const bf16* dy, const bf16* x, bf16* dx, long n8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8], g[8], o[8];
    ld8(x + i * 8, v);
    ld8(dy + i * 8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float x2 = v[j] * v[j];
        const float t = tanhf(0.7978845608028654f * (v[j] + 0.044715f * v[j] * x2));
        o[j] = g[j] * (0.5f * (1.f + t) + 0.5f * v[j] * (1.f - t * t) * 0.7978845608028654f * (1.f + 0.134145f * x2));
    }
    st8(dx + i * 8, o);
}

// ------------------------------------------------------------------------------------------------ cross entropy
__global__ __launch_bounds__(TPB) void cross_entropy_kernel(const bf16* logits, long ld, const int64_t* labels, float* loss_rows,
                                                            bf16* dlogits, long ld_d, int V, float inv_count) {
    // V need not be a multiple of 8 (a vocabulary grown by a few special tokens): rows are padded to ld >= ceil8(V) columns, the
    // pad columns are ignored on read (-inf) and their gradient is written as zero.
    __shared__ float red[8];
    const long row = blockIdx.x;
    const int64_t label = labels[row];
    const bf16* lr = logits + row * ld;
    const int nvec = (V + 7) / 8;
    // The target logit is read BEFORE any dlogits store: the engine calls this kernel in place (dlogits == logits), and other
    // waves of the block start overwriting the row as soon as they pass the barrier below.
    const float target = (label >= 0 && label < V) ? bf2f(lr[label]) : 0.f;
    if (label < 0 || label >= V) {  // ignore_index; labels >= V never index the row (the host side rejects them before launch)
        if (threadIdx.x == 0) loss_rows[row] = 0.f;
        if (dlogits) {
            const float z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int i = threadIdx.x; i < nvec; i += TPB) st8(dlogits + row * ld_d + i * 8, z);
        }
        return;
    }
    float m = -INFINITY, s = 0.f;
    for (int i = threadIdx.x; i < nvec; i += TPB) {
        float v[8];
        ld8(lr + i * 8, v);
        if (i * 8 + 8 > V) {
#pragma unroll
            for (int j = 0; j < 8; ++j) if (i * 8 + j >= V) v[j] = -INFINITY;
        }
        float vm = v[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) vm = fmaxf(vm, v[j]);
        const float mn = fmaxf(m, vm);
        float a = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) a += __expf(v[j] - mn);
        s = s * __expf(m - mn) + a;
        m = mn;
    }
    // block combine of (m, s)
    const float wm = wave_max(m);
    s = wave_sum(m == -INFINITY ? 0.f : s * __expf(m - wm));
    if (lane_id() == 0) { red[threadIdx.x >> 6] = wm; red[4 + (threadIdx.x >> 6)] = s; }
    __syncthreads();
    const float gm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float gs = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) gs += (red[i] == -INFINITY) ? 0.f : red[4 + i] * __expf(red[i] - gm);
    const float lse = gm + __logf(gs);
    if (threadIdx.x == 0) loss_rows[row] = lse - target;
    if (dlogits) {
        for (int i = threadIdx.x; i < nvec; i += TPB) {
            float v[8], o[8];
            ld8(lr + i * 8, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float p = __expf(v[j] - lse);
                o[j] = (i * 8 + j < V) ? (p - ((int64_t)(i * 8 + j) == label ? 1.f : 0.f)) * inv_count : 0.f;
            }
            st8(dlogits + row * ld_d + i * 8, o);
        }
    }
}

__global__ __launch_bounds__(1024) void sum_f32_kernel(const float* in, long n, float scale, float* out) {
    __shared__ float red[16];
    float s = 0.f;
    for (long i = threadIdx.x; i < n; i += 1024) s += in[i];
    s = wave_sum(s);
    if (lane_id() == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < 16; ++i) t += red[i];
        out[0] = t * scale;
    }
}

// ------------------------------------------------------------------------------------------------ gather / segment sum
__global__ void gather_rows_kernel(bf16* dst, long ld_dst, const bf16* ta, long ld_a, const bf16* tb, long ld_b, const int* idx, int d) {
    const long row = blockIdx.x;
    const int id = idx[row];
    const bf16* src = id >= 0 ? ta + (long)id * ld_a : (id == -1 ? nullptr : tb + (long)(-id - 2) * ld_b);
    for (int e = threadIdx.x * 8; e < d; e += blockDim.x * 8) {
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (src) v = *(const bf16x8*)(src + e);
        *(bf16x8*)(dst + row * ld_dst + e) = v;
    }
}

__global__ void segment_sum_rows_kernel(const bf16* src, long ld_src, const int* seg_off, const int* pos, const int* out_row,
                                        bf16* out, long ld_out, int d) {
    const int s = blockIdx.x;
    const int j0 = seg_off[s], j1 = seg_off[s + 1];
    bf16* o = out + (long)out_row[s] * ld_out;
    for (int e = threadIdx.x * 8; e < d; e += blockDim.x * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = j0; j < j1; ++j) {
            float v[8];
            ld8(src + (long)pos[j] * ld_src + e, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += v[i];
        }
        st8(o + e, acc);
    }
}

__global__ void weighted_segment_sum_rows_kernel(const bf16* src, long ld_src, const int* seg_off, const int* pos, const float* w,
                                                 const int* out_row, bf16* out, long ld_out, int d) {
    const int s = blockIdx.x;
    const int j0 = seg_off[s], j1 = seg_off[s + 1];
    bf16* o = out + (long)out_row[s] * ld_out;
    for (int e = threadIdx.x * 8; e < d; e += blockDim.x * 8) {
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int j = j0; j < j1; ++j) {
            float v[8];
            ld8(src + (long)pos[j] * ld_src + e, v);
            const float wj = w[j];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] += wj * v[i];
        }
        st8(o + e, acc);
    }
}

// out[out_row[s]] = elementwise max of the four rows src[idx4[s][0..3]]; which[s][e] = index (0..3) of the first maximum
// (nn.functional.max_pool2d's window scan order), kept for the backward pass.
__global__ void max4_rows_fwd_kernel(const bf16* src, long ld_src, const int* idx4, const int* out_row, bf16* out, long ld_out,
                                     unsigned char* which, int d) {
    const int s = blockIdx.x;
    bf16* o = out + (long)out_row[s] * ld_out;
    for (int e = threadIdx.x * 8; e < d; e += blockDim.x * 8) {
        float best[8];
        int arg[8];
        ld8(src + (long)idx4[s * 4] * ld_src + e, best);
#pragma unroll
        for (int i = 0; i < 8; ++i) arg[i] = 0;
        for (int j = 1; j < 4; ++j) {
            float v[8];
            ld8(src + (long)idx4[s * 4 + j] * ld_src + e, v);
#pragma unroll
            for (int i = 0; i < 8; ++i) if (v[i] > best[i] || v[i] != v[i]) { best[i] = v[i]; arg[i] = j; }
        }
        st8(o + e, best);
#pragma unroll
        for (int i = 0; i < 8; ++i) which[(long)s * d + e + i] = (unsigned char)arg[i];
    }
}
// dsrc[idx4[s][j]][e] = which[s][e] == j ? dout[dout_row[s]][e] : 0   (every source row belongs to exactly one window)
__global__ void max4_rows_bwd_kernel(const bf16* dout, long ld_dout, const int* idx4, const int* dout_row, const unsigned char* which,
                                     bf16* dsrc, long ld_dsrc, int d) {
    const int s = blockIdx.x;
    const bf16* g = dout + (long)dout_row[s] * ld_dout;
    for (int e = threadIdx.x * 8; e < d; e += blockDim.x * 8) {
        float gv[8];
        ld8(g + e, gv);
        for (int j = 0; j < 4; ++j) {
            float o[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] = which[(long)s * d + e + i] == j ? gv[i] : 0.f;
            st8(dsrc + (long)idx4[s * 4 + j] * ld_dsrc + e, o);
        }
    }
}

__global__ void add_pos_rows_kernel(bf16* x, const bf16* pos, long total, long per_image) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float a[8], b[8];
    ld8(x + i * 8, a);
    ld8(pos + (i % per_image) * 8, b);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += b[j];
    st8(x + i * 8, a);
}

// ------------------------------------------------------------------------------------------------ CLIP embeddings
__global__ void im2col_kernel(const bf16* pix, bf16* out, int n, int H, int W, int p, int Kp) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int gh = H / p, gw = W / p;
    const long total = (long)n * gh * gw * Kp;
    if (tid >= total) return;
    const int k = tid % Kp;
    const long row = tid / Kp;
    const int gx = row % gw, gy = (row / gw) % gh, img = row / ((long)gw * gh);
    bf16 v = f2bf(0.f);
    if (k < 3 * p * p) {
        const int c = k / (p * p), i = (k / p) % p, j = k % p;
        v = pix[(((long)img * 3 + c) * H + gy * p + i) * W + gx * p + j];
    }
    out[tid] = v;
}

__global__ void clip_embed_kernel(const bf16* patch_out, const bf16* cls, const bf16* pos, bf16* out, int n, int P, int d) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int per_row = d / 8;
    const long total = (long)n * (P + 1) * per_row;
    if (tid >= total) return;
    const int e = (tid % per_row) * 8;
    const long row = tid / per_row;
    const int t = row % (P + 1);
    const long img = row / (P + 1);
    float a[8], b[8], o[8];
    if (t == 0) ld8(cls + e, a); else ld8(patch_out + (img * P + t - 1) * d + e, a);
    ld8(pos + (long)t * d + e, b);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = a[i] + b[i];
    st8(out + row * d + e, o);
}

// ------------------------------------------------------------------------------------------------ image normalisation (device side)
// uint8 HWC canvases [n][gh*tile][gw*tile][3] -> bf16 CHW tiles [n*gh*gw][3][tile][tile], value = ((u8 rescaled) - mean[c]) / std[c] in the
// host processors' own fp32 arithmetic (mode 0: CLIPImageProcessor float32(u8) / 255; mode 1: SigLipImageProcessor float64(u8) * factor,
// then float32), rounded to bf16 -- bit-identical to "normalise on the host in fp32, cast on the device".
__global__ void normalize_tiles_u8_kernel(const unsigned char* img, bf16* out, int n, int gh, int gw, int tile, int mode, double factor,
                                          float m0, float m1, float m2, float s0, float s1, float s2) {
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per_tile = (long)tile * tile, total = (long)n * gh * gw * per_tile;
    if (tid >= total) return;
    const int x = tid % tile, y = (tid / tile) % tile;
    const long t = tid / per_tile;
    const int tx = t % gw, ty = (t / gw) % gh;
    const long im = t / ((long)gw * gh);
    const long W = (long)gw * tile;
    const unsigned char* p = img + ((im * gh * tile + (long)ty * tile + y) * W + (long)tx * tile + x) * 3;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = mode == 0 ? (float)p[c] / 255.0f : (float)((double)p[c] * factor);
        out[(t * 3 + c) * per_tile + (long)y * tile + x] = f2bf((v - mean[c]) / sd[c]);
    }
}

// ------------------------------------------------------------------------------------------------ transpose
// 64x64 tiles through LDS; out row c holds in[.., c] for r in [0, R_pad) (zeros beyond R).
__global__ __launch_bounds__(TPB) void transpose_kernel(const bf16* in, long in_ld, long in_bs0, long in_bs1, bf16* out, long out_ld,
                                                        long out_bs0, long out_bs1, int R, int C, int R_pad, int nb1, int perm32,
                                                        const int* cu) {
    __shared__ bf16 tile[64][72];
    const int bz = blockIdx.z, b0 = bz / nb1, b1 = bz % nb1;
    const bf16* src = in + (cu ? (long)cu[b0] * in_ld : b0 * in_bs0) + b1 * in_bs1;
    if (cu) R = cu[b0 + 1] - cu[b0];       // packed batches: this sample's own row count (zero padded up to R_pad)
    bf16* dst = out + b0 * out_bs0 + b1 * out_bs1;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ch = it * TPB + threadIdx.x;  // 512 chunks: 64 rows x 8 chunks
        const int r = ch >> 3, cc = (ch & 7) * 8;
        bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (r0 + r < R) {
            if (c0 + cc + 8 <= C) v = *(const bf16x8*)(src + (long)(r0 + r) * in_ld + c0 + cc);
            else
                for (int i = 0; i < 8; ++i) if (c0 + cc + i < C) v[i] = src[(long)(r0 + r) * in_ld + c0 + cc + i];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) tile[r][cc + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int ch = it * TPB + threadIdx.x;
        const int c = ch >> 3, rr = (ch & 7) * 8;
        if (c0 + c < C && r0 + rr < R_pad) {
            bf16x8 v;
            if (perm32) {
                // MFMA contraction order within each group of 32: output position 8g + 4h + j holds r = 16h + 4g + j
                const int q = ch & 7, base = (q >> 2) * 32 + (q & 3) * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = tile[base + (i >> 2) * 16 + (i & 3)][c];
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = tile[rr + i][c];
            }
            *(bf16x8*)(dst + (long)(c0 + c) * out_ld + r0 + rr) = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------ optimizer / misc
__global__ void adamw_kernel(bf16* p, float* master, const bf16* g, float* m, float* v, long n, float lr, float b1, float b2,
                             float eps, float wd, float bc1, float bc2, const float* gscale) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    const float gs = gscale ? gscale[0] : 1.f;
    const float rs2 = rsqrtf(bc2);
    const float step = lr / bc1;
    if (i + 4 <= n) {
        const bf16x4 gv = *(const bf16x4*)(g + i);
        f32x4 mv = *(f32x4*)(m + i), vv = *(f32x4*)(v + i), pv = *(f32x4*)(master + i);
        bf16x4 po;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float gg = bf2f(gv[j]) * gs;
            float pp = pv[j] * (1.f - lr * wd);
            mv[j] = b1 * mv[j] + (1.f - b1) * gg;
            vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
            pp -= step * mv[j] / (sqrtf(vv[j]) * rs2 + eps);
            pv[j] = pp;
            po[j] = f2bf(pp);
        }
        *(f32x4*)(m + i) = mv; *(f32x4*)(v + i) = vv; *(f32x4*)(master + i) = pv; *(bf16x4*)(p + i) = po;
    } else {
        for (long k = i; k < n; ++k) {
            const float gg = bf2f(g[k]) * gs;
            float pp = master[k] * (1.f - lr * wd);
            m[k] = b1 * m[k] + (1.f - b1) * gg;
            v[k] = b2 * v[k] + (1.f - b2) * gg * gg;
            pp -= step * m[k] / (sqrtf(v[k]) * rs2 + eps);
            master[k] = pp;
            p[k] = f2bf(pp);
        }
    }
}

__global__ __launch_bounds__(TPB) void sumsq_partial_kernel(const bf16* g, long n, float* partial) {
    __shared__ float red[4];
    float s = 0.f;
    const long n8 = n / 8;
    for (long i = (long)blockIdx.x * TPB + threadIdx.x; i < n8; i += (long)gridDim.x * TPB) {
        float v[8];
        ld8(g + i * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j] * v[j];
    }
    if (blockIdx.x == 0)
        for (long k = n8 * 8 + threadIdx.x; k < n; k += TPB) { const float t = bf2f(g[k]); s += t * t; }
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(TPB) void clip_coef_kernel(const float* partial, int nblk, float max_norm, float* out2) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblk; i += TPB) s += partial[i];
    s = block_sum<4>(s, red);
    if (threadIdx.x == 0) {
        const float norm = sqrtf(s);
        out2[0] = norm;
        out2[1] = fminf(1.f, max_norm / (norm + 1e-6f));
    }
}

__global__ void cast_f32_bf16_kernel(const float* in, bf16* out, long n) {
    // grid-stride: a dispatch carries a 32-bit work-item count, and the flat parameter buffers of a 7B model have 6.8e9 elements
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f2bf(in[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16* in, float* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = bf2f(in[i]);
}
__global__ void add_bf16_kernel(const bf16* a, const bf16* b, bf16* y, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        float x[8], z[8], o[8];
        ld8(a + i, x); ld8(b + i, z);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = x[j] + z[j];
        st8(y + i, o);
    } else {
        for (long k = i; k < n; ++k) y[k] = f2bf(bf2f(a[k]) + bf2f(b[k]));
    }
}

inline unsigned nblocks(long n, int per) { const long b = (n + per - 1) / per; return b >= (1L << 24) ? 0u : (unsigned)b; }   // 0 blocks = a launch error, loudly (see below)
// A dispatch's work-item count is a 32-bit field: blocks x threads must stay below 2^32 (a 7B model's flat buffers have 6.8e9 elements; a
// launch over one work-item per element silently covered n mod 2^32 of them).  Grid-stride kernels cap their grid with this; the
// others refuse such an n.
inline unsigned nblocks_capped(long n, int per) { const long b = (n + per - 1) / per; return (unsigned)(b > (1L << 22) ? (1L << 22) : b); }
inline bool fits_one_dispatch(long work_items) { return work_items < (1L << 32); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" const char* rv_version(void) { return "radvlm_hip 0.1 gfx950"; }

extern "C" int rv_rmsnorm_fwd(const void* x, const void* w, void* y, float* rstd, int rows, int d, float eps, void* stream) {
    if (!x || !w || !y || rows <= 0 || d <= 0 || (d & 7) || d > 8192) return RV_ERR_ARG;
    hipLaunchKernelGGL(rmsnorm_fwd_kernel, dim3(rows), dim3(TPB), 0, ST, (const bf16*)x, (const bf16*)w, (bf16*)y, rstd, d, eps);
    return rv_check_launch();
}
extern "C" int rv_rmsnorm_bwd(const void* dy, const void* x, const void* w, const float* rstd, void* dx, int dx_add,
                              float* dw_partial, int nblk, int rows, int d, void* stream) {
    if (!dy || !x || !w || !rstd || !dx || !dw_partial || nblk <= 0 || rows <= 0 || (d & 7) || d > 8192) return RV_ERR_ARG;
    hipLaunchKernelGGL(rmsnorm_bwd_kernel, dim3(nblk), dim3(TPB), 0, ST, (const bf16*)dy, (const bf16*)x, (const bf16*)w, rstd,
                       (bf16*)dx, dx_add, dw_partial, rows, d);
    return rv_check_launch();
}
extern "C" int rv_layernorm_fwd(const void* x, const void* w, const void* b, void* y, float* stats, int rows, int d, float eps, void* stream) {
    if (!x || !w || !b || !y || rows <= 0 || (d & 7) || d > 8192) return RV_ERR_ARG;
    hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(rows), dim3(TPB), 0, ST, (const bf16*)x, (const bf16*)w, (const bf16*)b, (bf16*)y, stats, d, eps);
    return rv_check_launch();
}
extern "C" int rv_layernorm_bwd(const void* dy, const void* x, const void* w, const float* stats, void* dx, int dx_add,
                                float* partial, int nblk, int rows, int d, void* stream) {
    if (!dy || !x || !w || !stats || !dx || !partial || nblk <= 0 || rows <= 0 || (d & 7) || d > 8192) return RV_ERR_ARG;
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(TPB), 0, ST, (const bf16*)dy, (const bf16*)x, (const bf16*)w, stats,
                       (bf16*)dx, dx_add, partial, rows, d);
    return rv_check_launch();
}
extern "C" int rv_quick_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(quick_gelu_fwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_quick_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(quick_gelu_bwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)x, (bf16*)dx, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_colsum_f32(const float* in, int rows, int cols, void* out, int accumulate, void* stream) {
    if (!in || !out || rows <= 0 || cols <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(nblocks(cols, 32)), dim3(1024), 0, ST, in, rows, cols, (bf16*)out, accumulate);
    return rv_check_launch();
}
extern "C" int rv_colsum_partial_bf16(const void* x, int64_t ld, int rows, int cols, float* partial, int nblk, void* stream) {
    if (!x || !partial || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || nblk <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblocks(cols, TPB * 8), nblk), dim3(TPB), 0, ST, (const bf16*)x, (long)ld, rows, cols, partial);
    return rv_check_launch();
}
extern "C" int rv_rope_inplace(void* x, int64_t ld, const float* cos_sin, int rows, int S, int heads, int hd, int nsec, int dir, void* stream) {
    if (!x || !cos_sin || rows <= 0 || S <= 0 || (hd & 15) || (ld & 7)) return RV_ERR_ARG;
    const long total = (long)rows * nsec * heads * (hd / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, (bf16*)x, (long)ld, cos_sin, (const int*)nullptr, rows, S, heads, hd, nsec, dir);
    return rv_check_launch();
}
extern "C" int rv_rope_inplace_pos(void* x, int64_t ld, const float* cos_sin, const int32_t* positions, int rows, int heads, int hd, int nsec, int dir, void* stream) {
    if (!x || !cos_sin || !positions || rows <= 0 || (hd & 15) || (ld & 7)) return RV_ERR_ARG;
    const long total = (long)rows * nsec * heads * (hd / 16);
    hipLaunchKernelGGL(rope_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, (bf16*)x, (long)ld, cos_sin, positions, rows, 1, heads, hd, nsec, dir);
    return rv_check_launch();
}
extern "C" int rv_swiglu_fwd(const void* gu, int64_t ld_gu, void* act, int64_t ld_act, int rows, int F, void* stream) {
    if (!gu || !act || rows <= 0 || (F & 7) || (ld_gu & 7) || (ld_act & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(nblocks((long)rows * (F / 8), 256)), dim3(256), 0, ST, (const bf16*)gu, (long)ld_gu, (bf16*)act, (long)ld_act, rows, F);
    return rv_check_launch();
}
extern "C" int rv_swiglu_bwd(const void* dact, int64_t ld_dact, const void* gu, int64_t ld_gu, void* dgu, int64_t ld_dgu, int rows, int F, void* stream) {
    if (!dact || !gu || !dgu || rows <= 0 || (F & 7) || (ld_gu & 7) || (ld_dact & 7) || (ld_dgu & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(nblocks((long)rows * (F / 8), 256)), dim3(256), 0, ST, (const bf16*)dact, (long)ld_dact, (const bf16*)gu, (long)ld_gu, (bf16*)dgu, (long)ld_dgu, rows, F);
    return rv_check_launch();
}
extern "C" int rv_dropout_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !y || n <= 0 || (n & 7) || p < 0.f || p >= 1.f) return RV_ERR_ARG;
    hipLaunchKernelGGL(dropout_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, (long)(n / 8), rv_dropout_thr16(p), 1.f / (1.f - p), (unsigned long long)seed);
    return rv_check_launch();
}
extern "C" int rv_dropout_add_bf16(const void* x, void* y, int64_t n, float p, uint64_t seed, void* stream) {
    if (!x || !y || n <= 0 || (n & 7) || p < 0.f || p >= 1.f) return RV_ERR_ARG;
    hipLaunchKernelGGL(dropout_add_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, (long)(n / 8), rv_dropout_thr16(p), 1.f / (1.f - p), (unsigned long long)seed);
    return rv_check_launch();
}
extern "C" int rv_gelu_fwd(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_gelu_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)x, (bf16*)dx, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_cross_entropy(const void* logits, int64_t ld, const int64_t* labels, float* loss_rows, void* dlogits, int64_t ld_d,
                                int rows, int V, float inv_count, void* stream) {
    if (!logits || !labels || !loss_rows || rows <= 0 || V <= 0 || (ld & 7) || ld < ((V + 7) & ~7) || (dlogits && ((ld_d & 7) || ld_d < ((V + 7) & ~7)))) return RV_ERR_ARG;
    hipLaunchKernelGGL(cross_entropy_kernel, dim3(rows), dim3(TPB), 0, ST, (const bf16*)logits, (long)ld, labels, loss_rows, (bf16*)dlogits, (long)ld_d, V, inv_count);
    return rv_check_launch();
}
extern "C" int rv_sum_f32(const float* in, int64_t n, float scale, float* out, void* stream) {
    if (!in || !out || n <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(sum_f32_kernel, dim3(1), dim3(1024), 0, ST, in, (long)n, scale, out);
    return rv_check_launch();
}
extern "C" int rv_gather_rows(void* dst, int64_t ld_dst, const void* ta, int64_t ld_a, const void* tb, int64_t ld_b, const int32_t* idx,
                              int rows, int d, void* stream) {
    if (!dst || !idx || rows <= 0 || (d & 7) || (ld_dst & 7) || (ld_a & 7) || (ld_b & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(rows), dim3(256), 0, ST, (bf16*)dst, (long)ld_dst, (const bf16*)ta, (long)ld_a, (const bf16*)tb, (long)ld_b, idx, d);
    return rv_check_launch();
}
extern "C" int rv_segment_sum_rows(const void* src, int64_t ld_src, const int32_t* seg_off, const int32_t* pos, const int32_t* out_row,
                                   int nseg, void* out, int64_t ld_out, int d, void* stream) {
    if (!src || !seg_off || !pos || !out_row || !out || nseg <= 0 || (d & 7) || (ld_src & 7) || (ld_out & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(segment_sum_rows_kernel, dim3(nseg), dim3(256), 0, ST, (const bf16*)src, (long)ld_src, seg_off, pos, out_row, (bf16*)out, (long)ld_out, d);
    return rv_check_launch();
}
extern "C" int rv_weighted_segment_sum_rows(const void* src, int64_t ld_src, const int32_t* seg_off, const int32_t* pos, const float* w,
                                            const int32_t* out_row, int nseg, void* out, int64_t ld_out, int d, void* stream) {
    if (!src || !seg_off || !pos || !w || !out_row || !out || nseg <= 0 || d <= 0 || (d & 7) || (ld_src & 7) || (ld_out & 7)) return RV_ERR_ARG;
    if ((((uintptr_t)src) | ((uintptr_t)out)) & 15) return RV_ERR_ARG;
    hipLaunchKernelGGL(weighted_segment_sum_rows_kernel, dim3(nseg), dim3(256), 0, ST, (const bf16*)src, (long)ld_src, seg_off, pos, w,
                       out_row, (bf16*)out, (long)ld_out, d);
    return rv_check_launch();
}
extern "C" int rv_max4_rows_fwd(const void* src, int64_t ld_src, const int32_t* idx4, const int32_t* out_row, int n, void* out, int64_t ld_out,
                                uint8_t* which, int d, void* stream) {
    if (!src || !idx4 || !out_row || !out || !which || n <= 0 || d <= 0 || (d & 7) || (ld_src & 7) || (ld_out & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(max4_rows_fwd_kernel, dim3(n), dim3(256), 0, ST, (const bf16*)src, (long)ld_src, idx4, out_row, (bf16*)out, (long)ld_out, which, d);
    return rv_check_launch();
}
extern "C" int rv_max4_rows_bwd(const void* dout, int64_t ld_dout, const int32_t* idx4, const int32_t* dout_row, int n, const uint8_t* which,
                                void* dsrc, int64_t ld_dsrc, int d, void* stream) {
    if (!dout || !idx4 || !dout_row || !which || !dsrc || n <= 0 || d <= 0 || (d & 7) || (ld_dout & 7) || (ld_dsrc & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(max4_rows_bwd_kernel, dim3(n), dim3(256), 0, ST, (const bf16*)dout, (long)ld_dout, idx4, dout_row, which, (bf16*)dsrc, (long)ld_dsrc, d);
    return rv_check_launch();
}
extern "C" int rv_add_pos_rows(void* x, const void* pos, int n, int P, int d, void* stream) {
    if (!x || !pos || n <= 0 || P <= 0 || d <= 0 || (d & 7) || ((((uintptr_t)x) | ((uintptr_t)pos)) & 15)) return RV_ERR_ARG;
    const long per = (long)P * d / 8, total = per * n;
    hipLaunchKernelGGL(add_pos_rows_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, (bf16*)x, (const bf16*)pos, total, per);
    return rv_check_launch();
}
extern "C" int rv_gelu_tanh_fwd(const void* x, void* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(gelu_tanh_fwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)x, (bf16*)y, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_gelu_tanh_bwd(const void* dy, const void* x, void* dx, int64_t n, void* stream) {
    if (!dy || !x || !dx || n <= 0 || (n & 7)) return RV_ERR_ARG;
    hipLaunchKernelGGL(gelu_tanh_bwd_kernel, dim3(nblocks(n / 8, 256)), dim3(256), 0, ST, (const bf16*)dy, (const bf16*)x, (bf16*)dx, (long)(n / 8));
    return rv_check_launch();
}
extern "C" int rv_im2col_patches(const void* pix, void* out, int n, int H, int W, int p, int Kp, void* stream) {
    if (!pix || !out || n <= 0 || p <= 0 || H < p || W < p || Kp < 3 * p * p) return RV_ERR_ARG;
    const long total = (long)n * (H / p) * (W / p) * Kp;
    hipLaunchKernelGGL(im2col_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, (const bf16*)pix, (bf16*)out, n, H, W, p, Kp);
    return rv_check_launch();
}
extern "C" int rv_normalize_tiles_u8(const uint8_t* img, void* out, int n, int gh, int gw, int tile, int mode, double factor, const float* mean3,
                                     const float* std3, void* stream) {
    if (!img || !out || !mean3 || !std3 || n <= 0 || gh <= 0 || gw <= 0 || tile <= 0 || (mode != 0 && mode != 1)) return RV_ERR_ARG;
    const long total = (long)n * gh * gw * tile * tile;
    hipLaunchKernelGGL(normalize_tiles_u8_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, img, (bf16*)out, n, gh, gw, tile, mode, factor,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return rv_check_launch();
}
extern "C" int rv_clip_embed(const void* patch_out, const void* cls, const void* pos, void* out, int n, int P, int d, void* stream) {
    if (!patch_out || !cls || !pos || !out || n <= 0 || P <= 0 || (d & 7)) return RV_ERR_ARG;
    const long total = (long)n * (P + 1) * (d / 8);
    hipLaunchKernelGGL(clip_embed_kernel, dim3(nblocks(total, 256)), dim3(256), 0, ST, (const bf16*)patch_out, (const bf16*)cls, (const bf16*)pos, (bf16*)out, n, P, d);
    return rv_check_launch();
}
extern "C" int rv_transpose_bf16(const void* in, int64_t in_ld, int64_t in_bs0, int64_t in_bs1, void* out, int64_t out_ld,
                                 int64_t out_bs0, int64_t out_bs1, int R, int C, int R_pad, int nb0, int nb1, int perm32,
                                 void* stream) {
    if (!in || !out || R <= 0 || C <= 0 || R_pad < R || (R_pad & 7) || nb0 <= 0 || nb1 <= 0) return RV_ERR_ARG;
    if ((in_ld & 7) || (out_ld & 7) || (in_bs0 & 7) || (in_bs1 & 7) || (out_bs0 & 7) || (out_bs1 & 7)) return RV_ERR_ARG;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15) return RV_ERR_ARG;
    if (perm32 && (R_pad & 63)) return RV_ERR_ARG;
    dim3 grid((R_pad + 63) / 64, (C + 63) / 64, nb0 * nb1);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(TPB), 0, ST, (const bf16*)in, (long)in_ld, (long)in_bs0, (long)in_bs1, (bf16*)out,
                       (long)out_ld, (long)out_bs0, (long)out_bs1, R, C, R_pad, nb1, perm32, (const int*)nullptr);
    return rv_check_launch();
}
extern "C" int rv_transpose_bf16_varlen(const void* in, int64_t in_ld, const int32_t* cu_rows, int64_t in_bs1, void* out, int64_t out_ld,
                                        int64_t out_bs0, int64_t out_bs1, int R_max, int C, int R_pad, int nb0, int nb1, int perm32,
                                        void* stream) {
    if (!in || !out || !cu_rows || R_max <= 0 || C <= 0 || R_pad < R_max || (R_pad & 7) || nb0 <= 0 || nb1 <= 0) return RV_ERR_ARG;
    if ((in_ld & 7) || (out_ld & 7) || (in_bs1 & 7) || (out_bs0 & 7) || (out_bs1 & 7)) return RV_ERR_ARG;
    if ((((uintptr_t)in) | ((uintptr_t)out)) & 15) return RV_ERR_ARG;
    if (perm32 && (R_pad & 63)) return RV_ERR_ARG;
    dim3 grid((R_pad + 63) / 64, (C + 63) / 64, nb0 * nb1);
    hipLaunchKernelGGL(transpose_kernel, grid, dim3(TPB), 0, ST, (const bf16*)in, (long)in_ld, 0L, (long)in_bs1, (bf16*)out,
                       (long)out_ld, (long)out_bs0, (long)out_bs1, R_max, C, R_pad, nb1, perm32, cu_rows);
    return rv_check_launch();
}
extern "C" int rv_adamw(void* p, float* master, const void* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps,
                        float wd, float bc1, float bc2, const float* gscale, void* stream) {
    if (!p || !master || !g || !m || !v || n <= 0) return RV_ERR_ARG;
    if ((((uintptr_t)p) | ((uintptr_t)g)) & 7) return RV_ERR_ARG;
    if ((((uintptr_t)master) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) return RV_ERR_ARG;
    // four elements per work-item; slices beyond 2^32 work-items (1.7e10 elements) go out as several dispatches
    const int64_t chunk = (int64_t)1 << 33;
    for (int64_t o = 0; o < n; o += chunk) {
        const int64_t c = n - o < chunk ? n - o : chunk;
        hipLaunchKernelGGL(adamw_kernel, dim3(nblocks((c + 3) / 4, 256)), dim3(256), 0, ST, (bf16*)p + o, master + o, (const bf16*)g + o, m + o, v + o, (long)c,
                           lr, b1, b2, eps, wd, bc1, bc2, gscale);
    }
    return rv_check_launch();
}
extern "C" int rv_sumsq_partial_bf16(const void* g, int64_t n, float* partial, int nblk, void* stream) {
    if (!g || !partial || n <= 0 || nblk <= 0 || (((uintptr_t)g) & 15)) return RV_ERR_ARG;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nblk), dim3(TPB), 0, ST, (const bf16*)g, (long)n, partial);
    return rv_check_launch();
}
extern "C" int rv_clip_coef(const float* partial, int nblk, float max_norm, float* out2, void* stream) {
    if (!partial || !out2 || nblk <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(TPB), 0, ST, partial, nblk, max_norm, out2);
    return rv_check_launch();
}
extern "C" int rv_cast_f32_to_bf16(const float* in, void* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(nblocks_capped(n, 256)), dim3(256), 0, ST, in, (bf16*)out, (long)n);
    return rv_check_launch();
}
extern "C" int rv_cast_bf16_to_f32(const void* in, float* out, int64_t n, void* stream) {
    if (!in || !out || n <= 0) return RV_ERR_ARG;
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(nblocks_capped(n, 256)), dim3(256), 0, ST, (const bf16*)in, out, (long)n);
    return rv_check_launch();
}
extern "C" int rv_add_bf16(const void* a, const void* b, void* y, int64_t n, void* stream) {
    if (!a || !b || !y || n <= 0 || !fits_one_dispatch((n + 7) / 8)) return RV_ERR_ARG;
    hipLaunchKernelGGL(add_bf16_kernel, dim3(nblocks((n + 7) / 8, 256)), dim3(256), 0, ST, (const bf16*)a, (const bf16*)b, (bf16*)y, (long)n);
    return rv_check_launch();
}
