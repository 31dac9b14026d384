// Degree-bucketed molecular graph kernels (GraphConvLayer / GraphPool / GraphGather2D / nonsat), gfx950.
//
// Batch layout (what ConvMol.agglomerate_mols produces, reference jova/feat/mol_graphs.py:272-383):
//   atoms  [N, F]            rows sorted degree-major (bucket d = atoms with d neighbours), molecule-minor
//   deg_slice int64 [11, 2]  (start, size) of every bucket, on the device
//   adj_d  int32 [n_d, d]    neighbour rows of the atoms of bucket d (d = 1..10)
// A workgroup owns a tile of 64 consecutive atoms of ONE bucket, so the per-degree weights are workgroup-uniform
// and the whole layer is a single launch; deg_slice is read on the device (no host sync).
#include "common.h"
#include "jova_hip.h"

#define TM 64
#define NB 11

struct GcParams {
    const float* w[21];
    const float* b[21];
    const int* adj[10];
};
struct GcGrads {
    float* dw[21];
    float* db[21];
};

__device__ __forceinline__ bool locate_tiles(const long long* __restrict__ deg_slice, int blk, int group, int& deg,
                                             long long& bstart, long long& bsize, int& tile0, int& ntile) {
    for (int d = 0; d < NB; ++d) {
        const long long size = deg_slice[2 * d + 1];
        const int tiles = (int)((size + TM - 1) / TM);
        const int groups = (tiles + group - 1) / group;
        if (blk < groups) {
            deg = d;
            bstart = deg_slice[2 * d];
            bsize = size;
            tile0 = blk * group;
            ntile = min(group, tiles - tile0);
            return true;
        }
        blk -= groups;
    }
    return false;
}

template <typename T> struct AccT { typedef float type; };
template <> struct AccT<long long> { typedef long long type; };

// As[r][0..kcols) <- sum over the `deg` neighbours of bucket-row (lrow0 + r) of x[nbr][:]  (zero padded)
template <typename TIn>
__device__ __forceinline__ void stage_nbr_sum(const TIn* __restrict__ x, int n_atoms, int f_in,
                                              const int* __restrict__ adj_d, int deg, long long lrow0, int rows,
                                              float* As, int KS, int kcols, int wave, int lane) {
    typedef typename AccT<TIn>::type acc_t;
#pragma unroll 2
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wave * 16 + rr;
        acc_t s0 = 0, s1 = 0;
        if (r < rows) {
            const int* arow = adj_d + (lrow0 + r) * deg;
            for (int n = 0; n < deg; ++n) {
                int j = arow[n];
                j = min(max(j, 0), n_atoms - 1);
                const TIn* xr = x + (long long)j * f_in;
                if (lane < f_in) s0 += xr[lane];
                if (lane + 64 < f_in) s1 += xr[lane + 64];
            }
        }
        if (lane < kcols) As[r * KS + lane] = (float)s0;
        if (lane + 64 < kcols) As[r * KS + lane + 64] = (float)s1;
    }
}

// As[r][0..kcols) <- x[row0 + r][:]  (zero padded rows/cols)
template <typename TIn>
__device__ __forceinline__ void stage_rows(const TIn* __restrict__ x, int f_in, long long row0, int rows, float* As,
                                           int KS, int kcols, int wave, int lane) {
#pragma unroll 4
    for (int rr = 0; rr < 16; ++rr) {
        const int r = wave * 16 + rr;
        float v0 = 0.f, v1 = 0.f;
        if (r < rows) {
            const TIn* xr = x + (row0 + r) * f_in;
            if (lane < f_in) v0 = (float)xr[lane];
            if (lane + 64 < f_in) v1 = (float)xr[lane + 64];
        }
        if (lane < kcols) As[r * KS + lane] = v0;
        if (lane + 64 < kcols) As[r * KS + lane + 64] = v1;
    }
}

// Ws[n][0..kcols) <- W[n0 + n][:]  for n in [0,64)  (zero padded)
__device__ __forceinline__ void stage_weight(const float* __restrict__ W, int f_in, int f_out, int n0, float* Ws,
                                             int KS, int kcols, int wave, int lane) {
    stage_rows<float>(W, f_in, n0, min(64, f_out - n0), Ws, KS, kcols, wave, lane);
}

// ------------------------------------------------------------------------------------------------ forward
template <typename TIn>
__global__ void __launch_bounds__(256)
graphconv_fwd_kernel(const TIn* __restrict__ x, int n_atoms, int f_in, int f_out,
                     const long long* __restrict__ deg_slice, GcParams P, int relu, float* __restrict__ out,
                     double* __restrict__ stats) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int deg, tile0, ntile;
    long long bstart, bsize;
    if (!locate_tiles(deg_slice, blockIdx.x, 1, deg, bstart, bsize, tile0, ntile)) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KP = (f_in + 1) & ~1;          // even K for the 2-deep MFMA step
    const int KS = KP | 1;                   // odd LDS row stride: conflict-free ds_read_b32 down a column
    float* As = smem;
    float* Ws = smem + TM * KS;
    const long long lrow0 = (long long)tile0 * TM;
    const int rows = (int)min((long long)TM, bsize - lrow0);
    const long long row0 = bstart + lrow0;
    const int n0 = blockIdx.y * 64;
    const int wr = wave >> 1, wc = wave & 1;
    const int w_rel = 2 * (deg - 1), w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    if (deg > 0) {
        stage_nbr_sum<TIn>(x, n_atoms, f_in, P.adj[deg - 1], deg, lrow0, rows, As, KS, KP, wave, lane);
        stage_weight(P.w[w_rel], f_in, f_out, n0, Ws, KS, KP, wave, lane);
        __syncthreads();
        mfma_tile_lds(As + 32 * wr * KS, KS, 1, Ws + 32 * wc * KS, 1, KS, KP, acc, lane);
        __syncthreads();
    }
    stage_rows<TIn>(x, f_in, row0, rows, As, KS, KP, wave, lane);
    stage_weight(P.w[w_self], f_in, f_out, n0, Ws, KS, KP, wave, lane);
    __syncthreads();
    mfma_tile_lds(As + 32 * wr * KS, KS, 1, Ws + 32 * wc * KS, 1, KS, KP, acc, lane);

    const int col = n0 + 32 * wc + (lane & 31);
    float bias = 0.f;
    if (col < f_out) {
        bias = P.b[w_self][col];
        if (deg > 0) bias += P.b[w_rel][col];
    }
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = 32 * wr + mfma32_row(r, lane);
        float v = acc[r] + bias;
        if (relu) v = fmaxf(v, 0.f);
        if (row < rows && col < f_out) {
            out[(row0 + row) * f_out + col] = v;
            s += v;
            q += v * v;
        }
    }
    if (stats) {      // per-column sum / sum of squares for the BatchNorm that follows (fp64 accumulators)
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32 && col < f_out) {
            atomicAdd(&stats[col], (double)s);
            atomicAdd(&stats[f_out + col], (double)q);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Per tile: dZ = dY * (Y > 0);  dW_rel += dZ^T S, dW_self += dZ^T X, db += colsum(dZ);
// NEED_DX: G = dZ W_rel -> gbuf, T = dZ W_self -> dx  (dx[i] += sum_{j in nbr(i)} G[j] is the second kernel).
template <typename TIn, bool NEED_DX>
__global__ void __launch_bounds__(256)
graphconv_bwd_kernel(const TIn* __restrict__ x, int n_atoms, int f_in, int f_out,
                     const long long* __restrict__ deg_slice, GcParams P, const float* __restrict__ dy,
                     const float* __restrict__ y, GcGrads Gd, float* __restrict__ gbuf, float* __restrict__ dx,
                     int group) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int deg, tile0, ntile;
    long long bstart, bsize;
    if (!locate_tiles(deg_slice, blockIdx.x, group, deg, bstart, bsize, tile0, ntile)) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int NT = (f_in + 31) >> 5;          // 32-wide column tiles of the f_in dimension
    const int KC = 32 * NT;
    const int KS = KC + 1;
    const int ZS = 65;
    float* Zs = smem;                          // [64 rows][65]  dZ tile
    float* As = Zs + TM * ZS;                  // [64 rows][KS]  S or X tile
    float* Ws = As + TM * KS;                  // [64 o][KS]     W_rel / W_self
    const int mt = wave & 1;
    const int nt0 = wave >> 1, nt1 = (wave >> 1) + 2;
    const int w_rel = 2 * (deg - 1), w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;

    f32x16 a_rel0, a_rel1, a_self0, a_self1;
#pragma unroll
    for (int r = 0; r < 16; ++r) a_rel0[r] = a_rel1[r] = a_self0[r] = a_self1[r] = 0.f;
    float db_acc = 0.f;

    for (int t = 0; t < ntile; ++t) {
        const long long lrow0 = (long long)(tile0 + t) * TM;
        const int rows = (int)min((long long)TM, bsize - lrow0);
        const long long row0 = bstart + lrow0;
        __syncthreads();                       // previous tile's LDS readers are done
        for (int rr = 0; rr < 16; ++rr) {      // dZ tile: wave -> 16 rows, lane -> output column
            const int r = wave * 16 + rr;
            float v = 0.f;
            if (r < rows && lane < f_out) {
                v = dy[(row0 + r) * f_out + lane];
                if (y && !(y[(row0 + r) * f_out + lane] > 0.f)) v = 0.f;
            }
            Zs[r * ZS + lane] = v;
        }
        if (deg > 0) {
            stage_nbr_sum<TIn>(x, n_atoms, f_in, P.adj[deg - 1], deg, lrow0, rows, As, KS, KC, wave, lane);
            if (NEED_DX) stage_weight(P.w[w_rel], f_in, f_out, 0, Ws, KS, KC, wave, lane);
        }
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int r = 0; r < TM; ++r) s += Zs[r * ZS + tid];
            db_acc += s;
        }
        if (deg > 0) {
            // dW_rel[o][k] += sum_row dZ[row][o] * S[row][k]
            if (nt0 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, As + 32 * nt0, KS, 1, TM, a_rel0, lane);
            if (nt1 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, As + 32 * nt1, KS, 1, TM, a_rel1, lane);
            if (NEED_DX) {
                // G[row][k] = sum_o dZ[row][o] * W_rel[o][k]
#pragma unroll 1
                for (int nt = nt0; nt < NT; nt += 2) {
                    f32x16 g;
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] = 0.f;
                    mfma_tile_lds(Zs + 32 * mt * ZS, ZS, 1, Ws + 32 * nt, KS, 1, 64, g, lane);
                    const int k = 32 * nt + (lane & 31);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * mt + mfma32_row(r, lane);
                        if (row < rows && k < f_in) gbuf[(row0 + row) * f_in + k] = g[r];
                    }
                }
            }
            __syncthreads();
        }
        stage_rows<TIn>(x, f_in, row0, rows, As, KS, KC, wave, lane);
        if (NEED_DX) stage_weight(P.w[w_self], f_in, f_out, 0, Ws, KS, KC, wave, lane);
        __syncthreads();
        if (nt0 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, As + 32 * nt0, KS, 1, TM, a_self0, lane);
        if (nt1 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, As + 32 * nt1, KS, 1, TM, a_self1, lane);
        if (NEED_DX) {
#pragma unroll 1
            for (int nt = nt0; nt < NT; nt += 2) {
                f32x16 g;
#pragma unroll
                for (int r = 0; r < 16; ++r) g[r] = 0.f;
                mfma_tile_lds(Zs + 32 * mt * ZS, ZS, 1, Ws + 32 * nt, KS, 1, 64, g, lane);
                const int k = 32 * nt + (lane & 31);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * mt + mfma32_row(r, lane);
                    if (row < rows && k < f_in) dx[(row0 + row) * f_in + k] = g[r];
                }
            }
        }
    }
    // flush: one atomic per weight element per workgroup (partials summed in registers over `group` tiles)
    if (tid < 64 && tid < f_out) {
        atomicAdd(&Gd.db[w_self][tid], db_acc);
        if (deg > 0) atomicAdd(&Gd.db[w_rel][tid], db_acc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * mt + mfma32_row(r, lane);
        if (o >= f_out) continue;
        const int k0 = 32 * nt0 + (lane & 31), k1 = 32 * nt1 + (lane & 31);
        if (nt0 < NT && k0 < f_in) {
            atomicAdd(&Gd.dw[w_self][o * f_in + k0], a_self0[r]);
            if (deg > 0) atomicAdd(&Gd.dw[w_rel][o * f_in + k0], a_rel0[r]);
        }
        if (nt1 < NT && k1 < f_in) {
            atomicAdd(&Gd.dw[w_self][o * f_in + k1], a_self1[r]);
            if (deg > 0) atomicAdd(&Gd.dw[w_rel][o * f_in + k1], a_rel1[r]);
        }
    }
}

// dx[i] (holds T_i) += sum over j with i in nbr(j) of G[j].  Symmetric adjacency: a gather over i's own list;
// otherwise a scatter with float atomics (dx must already hold T: it does, from the previous launch).
__global__ void __launch_bounds__(256)
graphconv_bwd_dx_kernel(int n_atoms, int f, const long long* __restrict__ deg_slice, GcParams P,
                        const float* __restrict__ gbuf, float* __restrict__ dx, const int* __restrict__ sym_flag) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_atoms) return;
    int deg = -1;
    long long lrow = 0;
    for (int d = 0; d < NB; ++d) {
        const long long s = deg_slice[2 * d], c = deg_slice[2 * d + 1];
        if (row >= s && row < s + c) { deg = d; lrow = row - s; }
    }
    if (deg <= 0) return;
    const int* arow = P.adj[deg - 1] + lrow * deg;
    const bool sym = sym_flag != nullptr && *sym_flag != 0;
    if (sym) {
        float s0 = 0.f, s1 = 0.f;
        for (int n = 0; n < deg; ++n) {
            int j = min(max(arow[n], 0), n_atoms - 1);
            const float* g = gbuf + (long long)j * f;
            if (lane < f) s0 += g[lane];
            if (lane + 64 < f) s1 += g[lane + 64];
        }
        if (lane < f) dx[row * f + lane] += s0;
        if (lane + 64 < f) dx[row * f + lane + 64] += s1;
    } else {
        const float* g = gbuf + row * f;
        for (int n = 0; n < deg; ++n) {
            int j = min(max(arow[n], 0), n_atoms - 1);
            if (lane < f) atomicAdd(&dx[(long long)j * f + lane], g[lane]);
            if (lane + 64 < f) atomicAdd(&dx[(long long)j * f + lane + 64], g[lane + 64]);
        }
    }
}

// flag <- 1 iff every neighbour list is duplicate-free, self-loop-free and i in nbr(j) whenever j in nbr(i)
__global__ void adj_symmetry_kernel(int n_atoms, const long long* __restrict__ deg_slice, GcParams P,
                                    int* __restrict__ flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_atoms) return;
    long long st[NB], sz[NB];
    for (int d = 0; d < NB; ++d) { st[d] = deg_slice[2 * d]; sz[d] = deg_slice[2 * d + 1]; }
    int deg = -1;
    long long lrow = 0;
    for (int d = 0; d < NB; ++d)
        if (i >= st[d] && i < st[d] + sz[d]) { deg = d; lrow = i - st[d]; }
    if (deg < 0) { atomicAnd(flag, 0); return; }
    if (deg == 0) return;
    const int* arow = P.adj[deg - 1] + lrow * deg;
    bool ok = true;
    for (int n = 0; n < deg && ok; ++n) {
        const int j = arow[n];
        if (j < 0 || j >= n_atoms || j == i) { ok = false; break; }
        for (int m = 0; m < n; ++m) ok = ok && (arow[m] != j);
        int dj = -1;
        long long lj = 0;
        for (int d = 0; d < NB; ++d)
            if (j >= st[d] && j < st[d] + sz[d]) { dj = d; lj = j - st[d]; }
        if (dj <= 0) { ok = false; break; }
        const int* jrow = P.adj[dj - 1] + lj * dj;
        bool found = false;
        for (int m = 0; m < dj; ++m) found = found || (jrow[m] == (int)i);
        ok = ok && found;
    }
    if (!ok) atomicAnd(flag, 0);
}

// ------------------------------------------------------------------------------------------------ GraphPool
// out[i] = max(x[i], max_n x[nbr_n(i)]); first maximal candidate wins (self, then adjacency order), arg saved.
__global__ void __launch_bounds__(256)
graphpool_fwd_kernel(const float* __restrict__ x, int n_atoms, int f, const long long* __restrict__ deg_slice,
                     GcParams P, float* __restrict__ out, unsigned char* __restrict__ arg) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_atoms) return;
    int deg = 0;
    long long lrow = 0;
    for (int d = 0; d < NB; ++d) {
        const long long s = deg_slice[2 * d], c = deg_slice[2 * d + 1];
        if (row >= s && row < s + c) { deg = d; lrow = row - s; }
    }
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    for (int c0 = 0; c0 < f; c0 += 64) {
        const int c = c0 + lane;
        if (c >= f) break;
        float best = x[row * f + c];
        int a = 0;
        for (int n = 0; n < deg; ++n) {
            const int j = min(max(arow[n], 0), n_atoms - 1);
            const float v = x[(long long)j * f + c];
            if (v > best) { best = v; a = n + 1; }
        }
        out[row * f + c] = best;
        arg[row * f + c] = (unsigned char)a;
    }
}

__global__ void __launch_bounds__(256)
graphpool_bwd_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int n_atoms, int f,
                     const long long* __restrict__ deg_slice, GcParams P, float* __restrict__ dx,
                     const int* __restrict__ sym_flag) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + wave;
    if (row >= n_atoms) return;
    long long st[NB], sz[NB];
    for (int d = 0; d < NB; ++d) { st[d] = deg_slice[2 * d]; sz[d] = deg_slice[2 * d + 1]; }
    int deg = 0;
    long long lrow = 0;
    for (int d = 0; d < NB; ++d)
        if (row >= st[d] && row < st[d] + sz[d]) { deg = d; lrow = row - st[d]; }
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    const bool sym = sym_flag != nullptr && *sym_flag != 0;
    if (sym) {
        // gather: atom `row` receives dy[j][c] from every neighbour j whose arg-max points back at it
        int pos[10];
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            pos[n] = -1;
            if (n < deg) {
                const int j = arow[n];
                int dj = 0;
                long long lj = 0;
                for (int d = 0; d < NB; ++d)
                    if (j >= st[d] && j < st[d] + sz[d]) { dj = d; lj = j - st[d]; }
                if (dj > 0) {
                    const int* jrow = P.adj[dj - 1] + lj * dj;
                    for (int m = dj - 1; m >= 0; --m)
                        if (jrow[m] == (int)row) pos[n] = m + 1;
                }
            }
        }
        for (int c0 = 0; c0 < f; c0 += 64) {
            const int c = c0 + lane;
            if (c >= f) break;
            float s = arg[row * f + c] == 0 ? dy[row * f + c] : 0.f;
#pragma unroll
            for (int n = 0; n < 10; ++n) {
                if (n < deg) {
                    const long long j = arow[n];
                    if ((int)arg[j * f + c] == pos[n]) s += dy[j * f + c];
                }
            }
            dx[row * f + c] = s;
        }
    } else {
        // scatter (dx zeroed by the host wrapper): route dy[row][c] to the arg-max source row
        for (int c0 = 0; c0 < f; c0 += 64) {
            const int c = c0 + lane;
            if (c >= f) break;
            const int a = arg[row * f + c];
            const long long tgt = a == 0 ? row : (long long)min(max(arow[a - 1], 0), n_atoms - 1);
            atomicAdd(&dx[tgt * f + c], dy[row * f + c]);
        }
    }
}

// ------------------------------------------------------------------------------------------------ GraphGather2D
// out[a][b][:] = x[mol_off[b] + a][:] if a < n_b else 0   (contiguous chunks of the degree-sorted atom tensor)
__global__ void gather2d_fwd_kernel(const float* __restrict__ x, const int* __restrict__ mol_off, int a_max, int bsz,
                                    int g, float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a_max * bsz * g;
    if (e >= total) return;
    const int c = (int)(e % g);
    const long long ab = e / g;
    const int b = (int)(ab % bsz), a = (int)(ab / bsz);
    const int o = mol_off[b], n = mol_off[b + 1] - o;
    out[e] = a < n ? x[(long long)(o + a) * g + c] : 0.f;
}

__global__ void gather2d_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ mol_off, int a_max,
                                    int bsz, int g, float* __restrict__ dx) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a_max * bsz * g;
    if (e >= total) return;
    const int c = (int)(e % g);
    const long long ab = e / g;
    const int b = (int)(ab % bsz), a = (int)(ab / bsz);
    const int o = mol_off[b], n = mol_off[b + 1] - o;
    if (a < n) dx[(long long)(o + a) * g + c] = dout[e];
}

// ------------------------------------------------------------------------------------------------ nonsat
// y <- x; repeat y' = (2 y^3 / 3 + x) / (y^2 + 1) until mean|y' - y| <= ep or the evaluation index exceeds
// max_iter (reference jova/nn/models.py:458-478).  The stop rule is a GLOBAL mean over the padded tensor, so the
// trip count is data dependent.  One persistent launch decides it on the device: every workgroup owns a fixed set
// of elements, publishes its |delta| partial (one write-through 8-byte store), arrives on a monotonic counter and
// polls it; after the match every workgroup sums all partials in the same order, so all take the same decision.
// Hand-off form: guide section 6 Guideline 16 / MICROARCH "Valid forms" table row 1 (one lane stores sc1 + agent
// atomic add; consumer polls with sc1 loads and reads the payload with sc1 loads only).  Spins are bounded.
// ctrl[0] = evaluations done, ctrl[1] = arrival counter, ctrl[2] = timeout flag.
#define NS_MAX_BLOCKS 512
#define NS_THREADS 1024
__global__ void __launch_bounds__(NS_THREADS)
nonsat_persistent_kernel(const float* __restrict__ x, float* __restrict__ ya, float* __restrict__ yb, long long n,
                         float ep, int max_iter, double* partial, int* ctrl, float* __restrict__ out,
                         float* __restrict__ dfac) {
    __shared__ double red[NS_THREADS / 64];
    __shared__ int s_flag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = gridDim.x;
    const long long stride = (long long)nb * NS_THREADS;
    const long long e0 = (long long)blockIdx.x * NS_THREADS + tid;
    int evals = 0;
    for (int it = 0;; ++it) {
        const float* yin = (it & 1) ? yb : ya;
        float* yout = (it & 1) ? ya : yb;
        double acc = 0.0;
        for (long long e = e0; e < n; e += stride) {
            const float xv = x[e];
            const float y = it == 0 ? xv : yin[e];
            const float yn = (2.f * (y * y * y) / 3.f + xv) / (y * y + 1.f);
            yout[e] = yn;
            acc += (double)fabsf(yn - y);
        }
        acc = wave_sum_d(acc);
        if (lane == 0) red[wave] = acc;
        __syncthreads();
        if (tid == 0) {
            double s = 0.0;
            for (int w = 0; w < NS_THREADS / 64; ++w) s += red[w];
            __hip_atomic_store(&partial[(size_t)(it & 1) * NS_MAX_BLOCKS + blockIdx.x], s, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int want = (it + 1) * nb;
            const unsigned long long t0 = wall_clock64();
            int ok = 1;
            while (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > 200000000ull) { ok = 0; break; }     // 2 s at 100 MHz: give up
            }
            s_flag = ok;
        }
        __syncthreads();
        if (!s_flag) {
            if (tid == 0) __hip_atomic_store(&ctrl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            evals = it + 1;
            break;
        }
        double p = 0.0;
        if (tid < nb)
            p = __hip_atomic_load(&partial[(size_t)(it & 1) * NS_MAX_BLOCKS + tid], __ATOMIC_RELAXED,
                                  __HIP_MEMORY_SCOPE_AGENT);
        p = wave_sum_d(p);
        __syncthreads();
        if (lane == 0) red[wave] = p;
        __syncthreads();
        double tot = 0.0;
        for (int w = 0; w < NS_THREADS / 64; ++w) tot += red[w];
        __syncthreads();
        const float mean = (float)(tot / (double)n);
        if (mean <= ep || it > max_iter) { evals = it + 1; break; }
    }
    if (blockIdx.x == 0 && tid == 0) ctrl[0] = evals;
    // evaluation evals-1 wrote (evals-1 even ? yb : ya); its input y_prev is x (evals == 1) or the other buffer
    const bool res_in_b = ((evals - 1) & 1) == 0;
    const float* yr = res_in_b ? yb : ya;
    const float* yp = evals == 1 ? x : (res_in_b ? ya : yb);
    for (long long e = e0; e < n; e += stride) {
        const float pv = yp[e];
        out[e] = yr[e];
        dfac[e] = 1.f / (pv * pv + 1.f);      // d y_final / d x (gradient flows through the last evaluation only)
    }
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                           float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        out[e] = a[e] * b[e];
}

// ================================================================================================ C ABI
static int fill_params(GcParams& P, const int32_t* const* adj, const float* const* w, const float* const* b) {
    for (int i = 0; i < 10; ++i) P.adj[i] = adj ? adj[i] : nullptr;
    for (int i = 0; i < 21; ++i) {
        P.w[i] = w ? w[i] : nullptr;
        P.b[i] = b ? b[i] : nullptr;
    }
    return 0;
}

extern "C" int jova_graphconv_fwd(const void* atoms, int atoms_is_i64, int64_t n_atoms, int f_in, int f_out,
                                  const int64_t* deg_slice, const int32_t* const* adj, const float* const* weight,
                                  const float* const* bias, int relu, float* out, double* bn_stats, void* stream) {
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31), "graphconv_fwd: n_atoms %lld out of range", (long long)n_atoms);
    JOVA_CHECK_ARG(f_in >= 1 && f_in <= 128, "graphconv_fwd: f_in %d not in [1,128]", f_in);
    JOVA_CHECK_ARG(f_out >= 1, "graphconv_fwd: f_out %d", f_out);
    JOVA_CHECK_ARG(atoms && deg_slice && adj && weight && bias && out, "graphconv_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (bn_stats) JOVA_CHECK_HIP(hipMemsetAsync(bn_stats, 0, sizeof(double) * 2 * f_out, s));
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, weight, bias);
    const int KS = ((f_in + 1) & ~1) | 1;
    const size_t lds = sizeof(float) * 2 * TM * KS;
    dim3 grid(cdiv(n_atoms, TM) + NB, cdiv(f_out, 64));
    if (atoms_is_i64)
        hipLaunchKernelGGL(graphconv_fwd_kernel<long long>, grid, dim3(256), lds, s, (const long long*)atoms,
                           (int)n_atoms, f_in, f_out, (const long long*)deg_slice, P, relu, out, bn_stats);
    else
        hipLaunchKernelGGL(graphconv_fwd_kernel<float>, grid, dim3(256), lds, s, (const float*)atoms, (int)n_atoms,
                           f_in, f_out, (const long long*)deg_slice, P, relu, out, bn_stats);
    JOVA_CHECK_LAUNCH("graphconv_fwd");
    return JOVA_OK;
}

extern "C" int jova_graphconv_bwd(const void* atoms, int atoms_is_i64, int64_t n_atoms, int f_in, int f_out,
                                  const int64_t* deg_slice, const int32_t* const* adj, const float* const* weight,
                                  const float* dy, const float* y_or_null, float* const* dweight, float* const* dbias,
                                  float* dx_or_null, float* gbuf_or_null, const int* sym_flag, void* stream) {
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31), "graphconv_bwd: n_atoms out of range");
    JOVA_CHECK_ARG(f_in >= 1 && f_in <= 128, "graphconv_bwd: f_in %d not in [1,128]", f_in);
    JOVA_CHECK_ARG(f_out >= 1 && f_out <= 64, "graphconv_bwd: f_out %d not in [1,64]", f_out);
    JOVA_CHECK_ARG(atoms && deg_slice && adj && dy && dweight && dbias, "graphconv_bwd: null pointer");
    JOVA_CHECK_ARG(!dx_or_null || (gbuf_or_null && weight && !atoms_is_i64), "graphconv_bwd: dx needs gbuf, weights, f32 atoms");
    if (n_atoms == 0) return JOVA_OK;
    hipStream_t s = (hipStream_t)stream;
    GcParams P;
    fill_params(P, adj, weight, nullptr);
    GcGrads G;
    for (int i = 0; i < 21; ++i) { G.dw[i] = dweight[i]; G.db[i] = dbias[i]; }
    const int NT = (f_in + 31) / 32, KS = 32 * NT + 1;
    const size_t lds = sizeof(float) * (TM * 65 + 2 * TM * KS);
    const int tiles = cdiv(n_atoms, TM);
    const int group = tiles > 1024 ? cdiv(tiles, 768) : 1;
    dim3 grid(cdiv(tiles, group) + NB);
    if (atoms_is_i64)
        hipLaunchKernelGGL((graphconv_bwd_kernel<long long, false>), grid, dim3(256), lds, s, (const long long*)atoms,
                           (int)n_atoms, f_in, f_out, (const long long*)deg_slice, P, dy, y_or_null, G, nullptr,
                           nullptr, group);
    else if (!dx_or_null)
        hipLaunchKernelGGL((graphconv_bwd_kernel<float, false>), grid, dim3(256), lds, s, (const float*)atoms,
                           (int)n_atoms, f_in, f_out, (const long long*)deg_slice, P, dy, y_or_null, G, nullptr,
                           nullptr, group);
    else
        hipLaunchKernelGGL((graphconv_bwd_kernel<float, true>), grid, dim3(256), lds, s, (const float*)atoms,
                           (int)n_atoms, f_in, f_out, (const long long*)deg_slice, P, dy, y_or_null, G, gbuf_or_null,
                           dx_or_null, group);
    JOVA_CHECK_LAUNCH("graphconv_bwd");
    if (dx_or_null) {
        hipLaunchKernelGGL(graphconv_bwd_dx_kernel, dim3(cdiv(n_atoms, 4)), dim3(256), 0, s, (int)n_atoms, f_in,
                           (const long long*)deg_slice, P, gbuf_or_null, dx_or_null, sym_flag);
        JOVA_CHECK_LAUNCH("graphconv_bwd_dx");
    }
    return JOVA_OK;
}

extern "C" int jova_adj_symmetry(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj, int* flag,
                                 void* stream) {
    JOVA_CHECK_ARG(deg_slice && adj && flag, "adj_symmetry: null pointer");
    hipStream_t s = (hipStream_t)stream;
    const int one = 1;
    JOVA_CHECK_HIP(hipMemcpyAsync(flag, &one, sizeof(int), hipMemcpyHostToDevice, s));
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    hipLaunchKernelGGL(adj_symmetry_kernel, dim3(cdiv(n_atoms, 256)), dim3(256), 0, s, (int)n_atoms,
                       (const long long*)deg_slice, P, flag);
    JOVA_CHECK_LAUNCH("adj_symmetry");
    return JOVA_OK;
}

extern "C" int jova_graphpool_fwd(const float* x, int64_t n_atoms, int f, const int64_t* deg_slice,
                                  const int32_t* const* adj, float* out, uint8_t* argmax, void* stream) {
    JOVA_CHECK_ARG(x && deg_slice && adj && out && argmax, "graphpool_fwd: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31) && f >= 1, "graphpool_fwd: bad shape");
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    hipLaunchKernelGGL(graphpool_fwd_kernel, dim3(cdiv(n_atoms, 4)), dim3(256), 0, (hipStream_t)stream, x,
                       (int)n_atoms, f, (const long long*)deg_slice, P, out, argmax);
    JOVA_CHECK_LAUNCH("graphpool_fwd");
    return JOVA_OK;
}

extern "C" int jova_graphpool_bwd(const float* dy, const uint8_t* argmax, int64_t n_atoms, int f,
                                  const int64_t* deg_slice, const int32_t* const* adj, float* dx, const int* sym_flag,
                                  int sym_flag_host_hint, void* stream) {
    JOVA_CHECK_ARG(dy && argmax && deg_slice && adj && dx, "graphpool_bwd: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31) && f >= 1, "graphpool_bwd: bad shape");
    if (n_atoms == 0) return JOVA_OK;
    hipStream_t s = (hipStream_t)stream;
    (void)sym_flag_host_hint;
    // the scatter path accumulates into dx; zeroing is harmless for the gather path (it overwrites)
    JOVA_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)n_atoms * f, s));
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    hipLaunchKernelGGL(graphpool_bwd_kernel, dim3(cdiv(n_atoms, 4)), dim3(256), 0, s, dy, argmax, (int)n_atoms, f,
                       (const long long*)deg_slice, P, dx, sym_flag);
    JOVA_CHECK_LAUNCH("graphpool_bwd");
    return JOVA_OK;
}

extern "C" int jova_gather2d_fwd(const float* x, const int32_t* mol_off, int a_max, int batch, int g, float* out,
                                 void* stream) {
    JOVA_CHECK_ARG(x && mol_off && out && a_max >= 0 && batch >= 1 && g >= 1, "gather2d_fwd: bad argument");
    const long long total = (long long)a_max * batch * g;
    if (total == 0) return JOVA_OK;
    hipLaunchKernelGGL(gather2d_fwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, mol_off,
                       a_max, batch, g, out);
    JOVA_CHECK_LAUNCH("gather2d_fwd");
    return JOVA_OK;
}

extern "C" int jova_gather2d_bwd(const float* dout, const int32_t* mol_off, int a_max, int batch, int g, float* dx,
                                 void* stream) {
    JOVA_CHECK_ARG(dout && mol_off && dx && a_max >= 0 && batch >= 1 && g >= 1, "gather2d_bwd: bad argument");
    const long long total = (long long)a_max * batch * g;
    if (total == 0) return JOVA_OK;
    hipLaunchKernelGGL(gather2d_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, mol_off,
                       a_max, batch, g, dx);
    JOVA_CHECK_LAUNCH("gather2d_bwd");
    return JOVA_OK;
}

extern "C" size_t jova_nonsat_workspace_bytes(int64_t n) {
    return 64 + sizeof(double) * 2 * NS_MAX_BLOCKS + sizeof(float) * 2 * (size_t)(n > 0 ? n : 0);
}

static int g_num_cu = 0;

extern "C" int jova_nonsat_fwd(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac,
                               int* n_eval_dev, void* workspace, void* stream) {
    JOVA_CHECK_ARG(x && out && dfac && n_eval_dev && workspace && n >= 0, "nonsat_fwd: bad argument");
    JOVA_CHECK_ARG(max_iter >= 0 && max_iter <= 100000, "nonsat_fwd: max_iter %d", max_iter);
    hipStream_t s = (hipStream_t)stream;
    if (g_num_cu == 0) {
        int dev = 0, cu = 0;
        JOVA_CHECK_HIP(hipGetDevice(&dev));
        JOVA_CHECK_HIP(hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev));
        g_num_cu = cu > 0 ? cu : 1;
    }
    // workspace: ctrl[16 ints] | partial[2][NS_MAX_BLOCKS] doubles | ya[n] | yb[n]
    int* ctrl = (int*)workspace;
    double* partial = (double*)((char*)workspace + 64);
    float* ya = (float*)((char*)workspace + 64 + sizeof(double) * 2 * NS_MAX_BLOCKS);
    float* yb = ya + n;
    JOVA_CHECK_HIP(hipMemsetAsync(workspace, 0, 64 + sizeof(double) * 2 * NS_MAX_BLOCKS, s));
    if (n == 0) return JOVA_OK;
    // one workgroup per CU at most: every workgroup is resident, which the in-kernel barrier needs
    int blocks = (int)((n + NS_THREADS - 1) / NS_THREADS);
    const int cap = g_num_cu < NS_MAX_BLOCKS ? g_num_cu : NS_MAX_BLOCKS;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(nonsat_persistent_kernel, dim3(blocks), dim3(NS_THREADS), 0, s, x, ya, yb, (long long)n, ep,
                       max_iter, partial, ctrl, out, dfac);
    JOVA_CHECK_LAUNCH("nonsat_persistent");
    JOVA_CHECK_HIP(hipMemcpyAsync(n_eval_dev, ctrl, 3 * sizeof(int), hipMemcpyDeviceToDevice, s));
    return JOVA_OK;
}

extern "C" int jova_mul(const float* a, const float* b, int64_t n, float* out, void* stream) {
    JOVA_CHECK_ARG(a && b && out && n >= 0, "mul: bad argument");
    if (n == 0) return JOVA_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, out);
    JOVA_CHECK_LAUNCH("mul");
    return JOVA_OK;
}
