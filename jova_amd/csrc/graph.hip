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
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include <utility>
#include <vector>

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

// ------------------------------------------------------------------------------------------------ tile staging
// Both A operands of a tile at once: Ss[r][:] = sum of the neighbour rows of atom r, Xs[r][:] = its own row.
// All loads are unconditional (indices / rows clamped, results zeroed by selects) so a wave keeps many rows in
// flight; the adjacency indices of a row-group are fetched before any feature row.
//
// Fast form: fp32 rows that are a multiple of 4 wide.  16 lanes (f <= 64) or 32 lanes (f <= 128) cover one row with
// float4 loads, so one wave instruction moves 4 (or 2) whole rows.
__device__ __forceinline__ void stage_tile_vec4(const float* __restrict__ x, int n_atoms, int f_in, int ldx,
                                                const int* __restrict__ adj_d, int deg, long long lrow0,
                                                long long row0, int rows, float* Ss, float* Xs, int KS, int kcols,
                                                int wave, int lane) {
    const int lsh = f_in <= 64 ? 4 : 5;
    const int rpi = 64 >> lsh;
    const int sub = lane >> lsh, c4 = (lane & ((1 << lsh) - 1)) * 4;
    const bool cok = c4 < f_in;
    const int cc = cok ? c4 : 0;
#pragma unroll 2
    for (int it = 0; it < 16; it += rpi) {
        const int r = wave * 16 + it + sub;
        const bool rok = r < rows;
        const int rc = rok ? r : rows - 1;
        const float4 self = *reinterpret_cast<const float4*>(x + (row0 + rc) * ldx + cc);
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        if (deg > 0) {
            const int* arow = adj_d + (lrow0 + rc) * deg;
            for (int n = 0; n < deg; n += 4) {
                int j[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) j[u] = min(max(arow[min(n + u, deg - 1)], 0), n_atoms - 1);
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (long long)j[u] * ldx + cc);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float m = (n + u < deg) ? 1.f : 0.f;
                    sum.x += m * v[u].x; sum.y += m * v[u].y; sum.z += m * v[u].z; sum.w += m * v[u].w;
                }
            }
        }
        const bool ok = rok && cok;
        if (c4 < kcols) {
            float* sp = Ss + r * KS + c4;
            float* xp = Xs + r * KS + c4;
            sp[0] = ok ? sum.x : 0.f; sp[1] = ok ? sum.y : 0.f; sp[2] = ok ? sum.z : 0.f; sp[3] = ok ? sum.w : 0.f;
            xp[0] = ok ? self.x : 0.f; xp[1] = ok ? self.y : 0.f; xp[2] = ok ? self.z : 0.f; xp[3] = ok ? self.w : 0.f;
        }
    }
}

// Generic form (int64 atoms of layer 1, or rows that are not 16-byte multiples): lane -> column (lane, lane + 64),
// four rows of the wave's 16 in flight together.
template <typename TIn>
__device__ __forceinline__ void stage_tile_generic(const TIn* __restrict__ x, int n_atoms, int f_in, int ldx,
                                                   const int* __restrict__ adj_d, int deg, long long lrow0,
                                                   long long row0, int rows, float* Ss, float* Xs, int KS, int kcols,
                                                   int wave, int lane) {
    typedef typename AccT<TIn>::type acc_t;
    const bool c0ok = lane < f_in, c1ok = lane + 64 < f_in;
    const int c0 = c0ok ? lane : 0, c1 = c1ok ? lane + 64 : 0;
    for (int rq = 0; rq < 16; rq += 4) {
        acc_t s0[4], s1[4];
        TIn x0[4], x1[4];
        int rc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = wave * 16 + rq + u;
            rc[u] = r < rows ? r : rows - 1;
            const TIn* xr = x + (row0 + rc[u]) * ldx;
            x0[u] = xr[c0];
            x1[u] = xr[c1];
            s0[u] = 0;
            s1[u] = 0;
        }
        for (int n = 0; n < deg; ++n) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) j[u] = min(max(adj_d[(lrow0 + rc[u]) * deg + n], 0), n_atoms - 1);
            TIn a0[4], a1[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const TIn* xr = x + (long long)j[u] * ldx;
                a0[u] = xr[c0];
                a1[u] = xr[c1];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { s0[u] += a0[u]; s1[u] += a1[u]; }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int r = wave * 16 + rq + u;
            const bool rok = r < rows;
            if (lane < kcols) {
                Ss[r * KS + lane] = (rok && c0ok) ? (float)s0[u] : 0.f;
                Xs[r * KS + lane] = (rok && c0ok) ? (float)x0[u] : 0.f;
            }
            if (lane + 64 < kcols) {
                Ss[r * KS + lane + 64] = (rok && c1ok) ? (float)s1[u] : 0.f;
                Xs[r * KS + lane + 64] = (rok && c1ok) ? (float)x1[u] : 0.f;
            }
        }
    }
}

template <typename TIn>
__device__ __forceinline__ void stage_tile(const TIn* __restrict__ x, int n_atoms, int f_in, int ldx,
                                           const int* __restrict__ adj_d, int deg, long long lrow0, long long row0,
                                           int rows, float* Ss, float* Xs, int KS, int kcols, int wave, int lane) {
    stage_tile_generic<TIn>(x, n_atoms, f_in, ldx, adj_d, deg, lrow0, row0, rows, Ss, Xs, KS, kcols, wave, lane);
}
template <>
__device__ __forceinline__ void stage_tile<float>(const float* __restrict__ x, int n_atoms, int f_in, int ldx,
                                                  const int* __restrict__ adj_d, int deg, long long lrow0,
                                                  long long row0, int rows, float* Ss, float* Xs, int KS, int kcols,
                                                  int wave, int lane) {
    // rows padded to a multiple of 4 floats (ldx) are read whole with 16-byte loads; the pad columns hold zeros
    if ((ldx & 3) == 0 && ((((unsigned long long)x) & 15) == 0))
        stage_tile_vec4(x, n_atoms, (f_in + 3) & ~3, ldx, adj_d, deg, lrow0, row0, rows, Ss, Xs, KS, kcols, wave, lane);
    else
        stage_tile_generic<float>(x, n_atoms, f_in, ldx, adj_d, deg, lrow0, row0, rows, Ss, Xs, KS, kcols, wave, lane);
}

// ------------------------------------------------------------------------------------------------ forward
// Weights are re-packed once per call into MFMA B-fragment order,
//   Wp[w][p][cb][lane] = W_w[32*cb + (lane & 31)][2*p + (lane >> 5)]   (zero padded),
// so the B operand of k-pair p is ONE coalesced 256-byte load per wave, served from L2 (the 21 matrices total
// < 0.5 MB) and prefetched 8 k-pairs ahead in registers: no LDS and no long-lived registers for the weights,
// which keeps the kernel at 4 workgroups per CU while tiles of other workgroups are in their gather phase.
__global__ void graphconv_pack_kernel(GcParams P, int f_in, int f_out, int kh, int ncb, float* __restrict__ wp) {
    const long long total = 21ll * kh * ncb * 64;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int lane = (int)(e & 63);
        long long q = e >> 6;
        const int cb = (int)(q % ncb);
        q /= ncb;
        const int p = (int)(q % kh), w = (int)(q / kh);
        const int col = 32 * cb + (lane & 31), k = 2 * p + (lane >> 5);
        wp[e] = (col < f_out && k < f_in) ? P.w[w][(long long)col * f_in + k] : 0.f;
    }
}

// int64 atom features (what the featuriser / collator delivers for layer 1) -> fp32 rows padded to `ld` floats
__global__ void atoms_to_f32_kernel(const long long* __restrict__ x, long long n, int f, int ld,
                                    float* __restrict__ out) {
    const long long total = n * ld;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const long long r = e / ld;
        const int c = (int)(e - r * ld);
        out[e] = c < f ? (float)x[r * f + c] : 0.f;
    }
}

template <typename TIn>
__global__ void __launch_bounds__(256, (sizeof(TIn) == 8 ? 3 : 4))
graphconv_fwd_kernel(const TIn* __restrict__ x, int n_atoms, int f_in, int ldx, int f_out,
                     const long long* __restrict__ deg_slice, GcParams P, const float* __restrict__ wp, int kh,
                     int ncb, int relu, float* __restrict__ out, double* __restrict__ stats, int group) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int deg, tile0, ntile;
    long long bstart, bsize;
    if (!locate_tiles(deg_slice, blockIdx.x, group, deg, bstart, bsize, tile0, ntile)) return;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int KC = 2 * kh;                   // staged columns (zero padded beyond f_in)
    const int KS = KC + 1;                   // odd LDS row stride: conflict-free ds_read_b32 down a column
    float* Ss = smem;
    float* Xs = smem + TM * KS;
    const int wr = wave >> 1, wc = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y * 2 + wc;      // 32-column block of this wave
    const int w_rel = 2 * (deg - 1), w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    const int col = 32 * cb + i;
    const bool colok = col < f_out && cb < ncb;
    const int cbc = cb < ncb ? cb : ncb - 1;
    const float* wrel = wp + (((long long)(deg > 0 ? w_rel : w_self) * kh) * ncb + cbc) * 64 + lane;
    const float* wself = wp + (((long long)w_self * kh) * ncb + cbc) * 64 + lane;
    const long long wstep = (long long)ncb * 64;
    float bias = 0.f;
    if (colok) {
        bias = P.b[w_self][col];
        if (deg > 0) bias += P.b[w_rel][col];
    }
    double s = 0.0, q = 0.0;

    for (int t = 0; t < ntile; ++t) {
        const long long lrow0 = (long long)(tile0 + t) * TM;
        const int rows = (int)min((long long)TM, bsize - lrow0);
        const long long row0 = bstart + lrow0;
        __syncthreads();                         // the previous tile's fragment reads are done
        stage_tile<TIn>(x, n_atoms, f_in, ldx, deg > 0 ? P.adj[deg - 1] : nullptr, deg, lrow0, row0, rows, Ss, Xs, KS,
                        KC, wave, lane);
        __syncthreads();
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const float* sp = Ss + (32 * wr + i) * KS + h;
        const float* xp = Xs + (32 * wr + i) * KS + h;
        // two passes (neighbour sums x W_rel, own rows x W_self) over kh k-pairs, 8 at a time: the 8 A values come
        // from LDS and the 8 B values from L2, both issued ahead of the 8 dependent MFMAs
        for (int pass = deg > 0 ? 0 : 1; pass < 2; ++pass) {
            const float* ap = pass == 0 ? sp : xp;
            const float* bp = pass == 0 ? wrel : wself;
            for (int p0 = 0; p0 < kh; p0 += 8) {
                float a[8], b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = min(p0 + u, kh - 1);
                    b[u] = bp[p * wstep];
                    a[u] = ap[2 * p];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float av = (p0 + u < kh) ? a[u] : 0.f;
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b[u], acc, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * wr + mfma32_row(r, lane);
            float v = acc[r] + bias;
            if (relu) v = fmaxf(v, 0.f);
            if (row < rows && colok) {
                out[(row0 + row) * f_out + col] = v;
                s += (double)v;
                q += (double)v * (double)v;
            }
        }
    }
    if (stats) {      // per-column sum / sum of squares for the BatchNorm that follows (fp64 accumulators)
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32 && colok) {
            atomicAdd(&stats[col], s);
            atomicAdd(&stats[f_out + col], q);
        }
    }
}

// ------------------------------------------------------------------------------------------------ forward, LDS-DMA form
// The fast form of the forward (fp32 rows of 64 or 80 floats, 16-byte aligned).  Nothing is staged through registers:
//   * gather: every row a tile needs (the 32 atoms' own rows and their neighbour rows) is copied HBM -> LDS by
//     global_load_lds_dwordx4, 16 lanes per 256-byte row, all rows of the tile in flight at once, no VGPRs held.
//     Rows land as they are in memory: plane p (0 = own rows, 1.. = neighbours), atom r -> slot s = 32 p + r; the
//     16-byte chunk c of a row sits at chunk position c ^ (s & 15), chosen on the SOURCE address (the LDS side of the
//     DMA is lane-linear), so the fragment reads below are bank-conflict free.  Rows of 80 floats keep floats 64..79
//     in a second image [slot][16] with chunk position c ^ ((s >> 2) & 3).
//   * MFMA: a wave owns 32 atoms x 32 output columns (2 waves per workgroup).  The k order of the contraction is
//     permuted so that lane (i, h) reads ONE ds_read_b128 per four MFMAs: group g, step s uses features 8g + 4h + s.
//     The neighbour sum is formed on the fly while reading the fragments, the weights of the workgroup's degree
//     stay in registers (re-packed in the same k order) for all of its tiles.
//   * pipeline: the LDS holds NPL planes.  A tile of degree d needs 1 + d of them; when two tiles fit (d = 1 with 4
//     planes, d <= 3 with 8) the rows of tile t+1 are in flight while tile t is in the MFMA, and the stores of tile
//     t-1 are issued behind them, so a wave waits for memory only at the top of a tile for requests that have had a
//     whole MFMA phase to complete.  The adjacency indices are always fetched one tile further ahead.  Otherwise
//     the next tile's rows are issued right after the MFMA phase, in front of this tile's stores.
//   * degrees > NPL - 1 take several gather rounds; the partial neighbour sums go through the MFMA one after another
//     (the product is linear in the sum).  High degrees are dispatched first so their longer tiles are not the tail.
#define GT 32

__device__ __forceinline__ void glds16(const float* g, float* l) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

// (start, size) of every bucket with ONE batch of scalar loads, then the workgroup's bucket by selects: the dependent
// load-compare-branch chain of a per-bucket loop costs a memory round trip per bucket (3 us of a workgroup's life).
__device__ __forceinline__ bool locate_tiles_batch(const long long* __restrict__ deg_slice, int blk, int group, int& deg,
                                                   long long& bstart, int& bsize, int& tile0, int& ntile) {
    long long ds[2 * NB];
#pragma unroll
    for (int k = 0; k < 2 * NB; ++k) ds[k] = deg_slice[k];
    bool found = false;
    deg = 0; bstart = 0; bsize = 0; tile0 = 0; ntile = 0;
#pragma unroll
    for (int d = NB - 1; d >= 0; --d) {
        const int size = (int)ds[2 * d + 1];
        const int tiles = (size + GT - 1) / GT;
        const int groups = (tiles + group - 1) / group;
        const bool hit = !found && blk < groups;
        if (hit) {
            deg = d;
            bstart = ds[2 * d];
            bsize = size;
            tile0 = blk * group;
            ntile = min(group, tiles - tile0);
        }
        found = found || hit;
        blk -= groups;
    }
    return found;
}

// Work split decided on the host when it knows the bucket sizes (deg_slice_host): workgroups of bucket d are
// blocks [first[d], first[d] + count), each owns tpw[d] consecutive tiles, sized so that every workgroup has about
// the same number of steps and all of them are resident at once (one prologue / flush per workgroup slot).
struct BwdPlan {
    int use;
    int first[NB + 1];       // first[d] .. first[d - 1] - 1 are the blocks of bucket d (buckets in descending order)
    int tpw[NB];
    int start[NB];
    int size[NB];
};

__device__ __forceinline__ bool locate_tiles_plan(const BwdPlan& pl, int blk, int& deg, long long& bstart, int& bsize,
                                                  int& tile0, int& ntile) {
    bool found = false;
    deg = 0; bstart = 0; bsize = 0; tile0 = 0; ntile = 0;
#pragma unroll
    for (int d = NB - 1; d >= 0; --d) {
        const int b0 = pl.first[d + 1], b1 = pl.first[d];          // first[NB] = 0
        if (blk >= b0 && blk < b1) {
            found = true;
            deg = d;
            bstart = pl.start[d];
            bsize = pl.size[d];
            tile0 = (blk - b0) * pl.tpw[d];
            ntile = min(pl.tpw[d], (bsize + GT - 1) / GT - tile0);
        }
    }
    return found && ntile > 0;
}

// weights -> B fragments in the permuted k order:
//   Wq[w][cb][g][lane][s] = W_w[32 cb + (lane & 31)][8 g + 4 (lane >> 5) + s]   (zero padded)
__global__ void graphconv_pack_dma_kernel(GcParams P, int f_in, int f_out, int kg, int ncb, float* __restrict__ wq) {
    const long long total = 21ll * ncb * kg * 256;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total;
         e += (long long)gridDim.x * blockDim.x) {
        const int s = (int)(e & 3), lane = (int)((e >> 2) & 63);
        long long q = e >> 8;
        const int g = (int)(q % kg);
        q /= kg;
        const int cb = (int)(q % ncb), w = (int)(q / ncb);
        const int col = 32 * cb + (lane & 31), k = 8 * g + 4 * (lane >> 5) + s;
        wq[e] = (col < f_out && k < f_in) ? P.w[w][(long long)col * f_in + k] : 0.f;
    }
}

// Where the planes live.  The LDS is two separate static arrays of NPL/2 planes each (main image, then tail image):
// while tile t is read from one array the DMA of tile t+1 writes the other, and because they are distinct objects the
// compiler does not put a vmcnt(0) in front of the fragment reads (it does when DMA and reads share one array).
template <int KG, int NPL>
struct HalfBuf {                                                 // planes 0 .. NPL/2-1 of ONE array
    float* base;
    __device__ __forceinline__ float* main(int lp) const { return base + lp * (GT * 64); }
    __device__ __forceinline__ float* tail(int lp) const { return base + (NPL / 2) * (GT * 64) + lp * (GT * 16); }
};
template <int KG, int NPL>
struct FullBuf {                                                 // planes 0 .. NPL-1 across both arrays
    float* b0;
    float* b1;
    __device__ __forceinline__ float* main(int pp) const {
        return (pp < NPL / 2 ? b0 : b1) + (pp % (NPL / 2)) * (GT * 64);
    }
    __device__ __forceinline__ float* tail(int pp) const {
        return (pp < NPL / 2 ? b0 : b1) + (NPL / 2) * (GT * 64) + (pp % (NPL / 2)) * (GT * 16);
    }
};

// Row numbers of the planes [p_begin, p_end) of a tile (plane 0 = own rows, plane p >= 1 = neighbour nb0 + p - 1),
// in the two lane -> slot maps the DMA uses (main image: 16 lanes per slot; tail image: 4 lanes per slot).
template <int KG, int NPL>
struct RowIdx {
    int main[NPL * 4];
    int tail[KG > 8 ? NPL : 1];
    // (no arithmetic on the loaded values here: while a DMA is in flight a use of an ordinary load's result makes
    // the compiler wait for vmcnt(0), which would drain the DMA too; they are clamped at issue time)
    __device__ __forceinline__ void load(const int* __restrict__ adj_d, int deg, int lrow0, long long row0, int rows,
                                         int nb0, int p_begin, int p_end, int tid) {
        const int sub = tid >> 4;
#pragma unroll
        for (int it = 0; it < NPL * 4; ++it) {
            const int p = it >> 2;
            const int r = (it & 3) * 8 + sub;
            const int rc = r < rows ? r : rows - 1;
            main[it] = 0;
            if (p >= p_begin && p < p_end) {
                if (p == 0) main[it] = (int)(row0 + rc);
                else main[it] = adj_d[__umul24((unsigned)(lrow0 + rc), (unsigned)deg) + (unsigned)(nb0 + p - 1)];
            }
        }
        if (KG > 8) {
            const int r = tid >> 2;
            const int rc = r < rows ? r : rows - 1;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                tail[p] = 0;
                if (p >= p_begin && p < p_end) {
                    if (p == 0) tail[p] = (int)(row0 + rc);
                    else tail[p] = adj_d[__umul24((unsigned)(lrow0 + rc), (unsigned)deg) + (unsigned)(nb0 + p - 1)];
                }
            }
        }
    }
    // issues the DMA of those planes into the planes of `buf`
    template <typename Buf>
    __device__ __forceinline__ void issue(const float* __restrict__ x, int n_atoms, int ldx, int p_begin, int p_end,
                                          const Buf& buf, int tid) const {
        const int wave = tid >> 6;
        {
            const int sub = tid >> 4, pos = tid & 15;
#pragma unroll
            for (int it = 0; it < NPL * 4; ++it) {
                const int p = it >> 2;
                const int sw = ((it & 1) * 8 + sub) & 15;        // slot & 15
                if (p >= p_begin && p < p_end)
                    glds16(x + (__umul24(min((unsigned)main[it], (unsigned)(n_atoms - 1)), (unsigned)ldx) + ((pos ^ sw) << 2)),
                           buf.main(p) + ((it & 3) * 8 + wave * 4) * 64);
            }
        }
        if (KG > 8) {
            const int r = tid >> 2, pos = tid & 3;
#pragma unroll
            for (int p = 0; p < NPL; ++p) {
                if (p >= p_begin && p < p_end)
                    glds16(x + (__umul24(min((unsigned)tail[p], (unsigned)(n_atoms - 1)), (unsigned)ldx) + 64 + ((pos ^ ((r >> 2) & 3)) << 2)),
                           buf.tail(p) + (wave * 16) * 16);
            }
        }
    }
};

template <int G, typename Buf>
__device__ __forceinline__ float4 frag_dma(const Buf& buf, int p, int r, int h) {
    if (G < 8) return *reinterpret_cast<const float4*>(buf.main(p) + r * 64 + (((2 * G + h) ^ (r & 15)) << 2));
    return *reinterpret_cast<const float4*>(buf.tail(p) + r * 16 + (((2 * (G - 8) + h) ^ ((r >> 2) & 3)) << 2));
}

// acc += (sum of planes p0 .. p0+cnt-1)[32 atoms x 8 KG] * b   in the permuted k order
template <int KG, int G>
struct DmaPass {
    template <typename Buf>
    static __device__ __forceinline__ void run(const Buf& buf, int p0, int cnt, int r, int h, const float4* b,
                                               f32x16& acc) {
        float4 a = frag_dma<G>(buf, p0, r, h);
        for (int n = 1; n < cnt; ++n) {
            const float4 v = frag_dma<G>(buf, p0 + n, r, h);
            a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[G].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[G].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[G].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[G].w, acc, 0, 0, 0);
        DmaPass<KG, G + 1>::run(buf, p0, cnt, r, h, b, acc);
    }
};
template <int KG>
struct DmaPass<KG, KG> {
    template <typename Buf>
    static __device__ __forceinline__ void run(const Buf&, int, int, int, int, const float4*, f32x16&) {}
};

#ifndef JOVA_FWD_WPE
#define JOVA_FWD_WPE 2
#endif
template <int KG, int NPL>
__global__ void __launch_bounds__(128, (NPL <= 4 ? JOVA_FWD_WPE : 1))
graphconv_fwd_dma_kernel(const float* __restrict__ x, int n_atoms, int ldx, int f_out,
                         const long long* __restrict__ deg_slice, GcParams P, const float* __restrict__ wq, int ncb,
                         int relu, float* __restrict__ out, double* __restrict__ stats, int group, BwdPlan plan) {
    constexpr int HALF = (NPL / 2) * GT * (KG > 8 ? 80 : 64);
    __shared__ __attribute__((aligned(16))) float lds0[HALF];
    __shared__ __attribute__((aligned(16))) float lds1[HALF];
    int deg, tile0, ntile, bsize;
    long long bstart;
    if (plan.use) {
        if (!locate_tiles_plan(plan, blockIdx.x, deg, bstart, bsize, tile0, ntile)) return;
    } else if (!locate_tiles_batch(deg_slice, blockIdx.x, group, deg, bstart, bsize, tile0, ntile)) {
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int cb = blockIdx.y * 2 + wave;
    const int col = 32 * cb + i;
    const bool colok = col < f_out && cb < ncb;
    const int cbc = cb < ncb ? cb : ncb - 1;
    const int w_rel = deg > 0 ? 2 * (deg - 1) : 20, w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    const int* adj_d = deg > 0 ? P.adj[deg - 1] : nullptr;
    const int cnt0 = min(deg, NPL - 1);          // neighbour planes of the first round
    const int ppt = cnt0 + 1;                    // planes a tile occupies
    const bool db = 2 * ppt <= NPL;              // a tile fits one array: software pipeline over the two
    const HalfBuf<KG, NPL> half0{lds0}, half1{lds1};
    const FullBuf<KG, NPL> full{lds0, lds1};

    RowIdx<KG, NPL> idx;
    idx.load(adj_d, deg, tile0 * GT, bstart + (long long)tile0 * GT, min(GT, bsize - tile0 * GT), 0, 0, ppt, tid);
    float4 bself[KG], brel[KG];
    {
        const float4* ws = reinterpret_cast<const float4*>(wq) + (((long long)w_self * ncb + cbc) * KG) * 64 + lane;
        const float4* wr = reinterpret_cast<const float4*>(wq) + (((long long)w_rel * ncb + cbc) * KG) * 64 + lane;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            bself[g] = ws[g * 64];
            brel[g] = wr[g * 64];
        }
    }
    float bias = 0.f;
    if (colok) {
        bias = P.b[w_self][col];
        if (deg > 0) bias += P.b[w_rel][col];
    }
    double s = 0.0, q = 0.0;
    if (db) idx.issue(x, n_atoms, ldx, 0, ppt, half0, tid);
    else idx.issue(x, n_atoms, ldx, 0, ppt, full, tid);
    if (ntile > 1)
        idx.load(adj_d, deg, (tile0 + 1) * GT, bstart + (long long)(tile0 + 1) * GT, min(GT, bsize - (tile0 + 1) * GT),
                 0, 0, ppt, tid);

    // (wave-uniform base + 32-bit lane offsets; full tiles store without per-row tests; fp64 statistics only on request)
    auto store_tile = [&](const f32x16& a, long long row0, int rows) {
        if (!colok) return;
        float* ob = out + row0 * f_out;
        const unsigned fo = (unsigned)f_out, co = (unsigned)col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned row = (unsigned)mfma32_row(r, lane);
            float v = a[r] + bias;
            if (relu) v = fmaxf(v, 0.f);
            if (rows == GT || (int)row < rows) {
                ob[row * fo + co] = v;
                if (stats) {
                    s += (double)v;
                    q += (double)v * (double)v;
                }
            }
        }
    };
    auto prefetch_idx = [&](int t) {             // adjacency indices of tile t (if the workgroup has one)
        if (t < ntile) {
            const int l = (tile0 + t) * GT;
            idx.load(adj_d, deg, l, bstart + l, min(GT, bsize - l), 0, 0, ppt, tid);
        }
    };

    if (db) {
        f32x16 acc, prev;
#pragma unroll
        for (int r = 0; r < 16; ++r) prev[r] = 0.f;
        long long prev_row0 = 0;
        int prev_rows = 0;
        auto step = [&](const HalfBuf<KG, NPL>& cur, const HalfBuf<KG, NPL>& nxt, int t) {
            const int lrow0 = (tile0 + t) * GT;
            // rows of tile t landed (and the prefetched indices, and the stores of tile t-2: all issued a whole MFMA
            // phase ago); every wave is past its reads of tile t-1.  The wait is explicit: with the two arrays being
            // distinct objects the barrier alone does not drain the DMA.
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
            __syncthreads();
            if (t + 1 < ntile) idx.issue(x, n_atoms, ldx, 0, ppt, nxt, tid);
            prefetch_idx(t + 2);
            if (t > 0) store_tile(prev, prev_row0, prev_rows);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            DmaPass<KG, 0>::run(cur, 0, 1, i, h, bself, acc);
            if (cnt0 > 0) DmaPass<KG, 0>::run(cur, 1, cnt0, i, h, brel, acc);
            prev = acc;
            prev_row0 = bstart + lrow0;
            prev_rows = min(GT, bsize - lrow0);
        };
        for (int t = 0; t < ntile; t += 2) {
            step(half0, half1, t);
            if (t + 1 < ntile) step(half1, half0, t + 1);
        }
        store_tile(prev, prev_row0, prev_rows);
    } else {
        for (int t = 0; t < ntile; ++t) {
            const int lrow0 = (tile0 + t) * GT;
            const int rows = min(GT, bsize - lrow0);
            const long long row0 = bstart + lrow0;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): rows of tile t (first round) landed
            __syncthreads();
            DmaPass<KG, 0>::run(full, 0, 1, i, h, bself, acc);
            if (cnt0 > 0) DmaPass<KG, 0>::run(full, 1, cnt0, i, h, brel, acc);
            for (int nb0 = NPL - 1; nb0 < deg; nb0 += NPL - 1) {  // further rounds of a high-degree tile
                const int cnt = min(NPL - 1, deg - nb0);
                RowIdx<KG, NPL> more;
                more.load(adj_d, deg, lrow0, row0, rows, nb0, 1, cnt + 1, tid);
                __syncthreads();
                more.issue(x, n_atoms, ldx, 1, cnt + 1, full, tid);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                __syncthreads();
                DmaPass<KG, 0>::run(full, 1, cnt, i, h, brel, acc);
            }
            __syncthreads();                     // this tile's fragment reads are done
            if (t + 1 < ntile) idx.issue(x, n_atoms, ldx, 0, ppt, full, tid);
            prefetch_idx(t + 2);
            store_tile(acc, row0, rows);
        }
    }
    if (stats) {
        s += __shfl_xor(s, 32, 64);
        q += __shfl_xor(q, 32, 64);
        if (lane < 32 && colok) {
            atomicAdd(&stats[col], s);
            atomicAdd(&stats[f_out + col], q);
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward
// Per tile: dZ = dY * (Y > 0);  dW_rel += dZ^T S, dW_self += dZ^T X, db += colsum(dZ)  (accumulated in registers over
// the workgroup's `group` tiles, one atomic flush at the end);
// NEED_DX: G = dZ W_rel -> gbuf, T = dZ W_self -> dx  (dx[i] += sum_{j in nbr(i)} G[j] is the second kernel);
// the B operand of those two products is read straight from the weight matrices (row o, 32 consecutive columns:
// coalesced, L2 resident).
template <typename TIn, bool NEED_DX>
__global__ void __launch_bounds__(256, 2)
graphconv_bwd_kernel(const TIn* __restrict__ x, int n_atoms, int f_in, int ldx, int f_out,
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
    float* Ss = Zs + TM * ZS;                  // [64 rows][KS]  neighbour sums
    float* Xs = Ss + TM * KS;                  // [64 rows][KS]  own rows
    const int mt = wave & 1;
    const int nt0 = wave >> 1, nt1 = (wave >> 1) + 2;
    const int i = lane & 31, h = lane >> 5;
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
#pragma unroll 4
        for (int rr = 0; rr < 16; ++rr) {      // dZ tile: wave -> 16 rows, lane -> output column (unconditional loads)
            const int r = wave * 16 + rr;
            const bool ok = r < rows && lane < f_out;
            const long long off = (row0 + (r < rows ? r : rows - 1)) * f_out + (lane < f_out ? lane : 0);
            float v = dy[off];
            if (y) v = y[off] > 0.f ? v : 0.f;
            Zs[r * ZS + lane] = ok ? v : 0.f;
        }
        stage_tile<TIn>(x, n_atoms, f_in, ldx, deg > 0 ? P.adj[deg - 1] : nullptr, deg, lrow0, row0, rows, Ss, Xs, KS,
                        KC, wave, lane);
        __syncthreads();
        if (tid < 64) {
            float s = 0.f;
            for (int r = 0; r < TM; ++r) s += Zs[r * ZS + tid];
            db_acc += s;
        }
        // dW[o][k] += sum_row dZ[row][o] * A[row][k]
        if (deg > 0) {
            if (nt0 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, Ss + 32 * nt0, KS, 1, TM, a_rel0, lane);
            if (nt1 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, Ss + 32 * nt1, KS, 1, TM, a_rel1, lane);
        }
        if (nt0 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, Xs + 32 * nt0, KS, 1, TM, a_self0, lane);
        if (nt1 < NT) mfma_tile_lds(Zs + 32 * mt, 1, ZS, Xs + 32 * nt1, KS, 1, TM, a_self1, lane);
        if (NEED_DX) {
            // G[row][k] = sum_o dZ[row][o] W_rel[o][k];  T[row][k] = sum_o dZ[row][o] W_self[o][k]
#pragma unroll 1
            for (int pass = deg > 0 ? 0 : 1; pass < 2; ++pass) {
                const float* W = P.w[pass == 0 ? w_rel : w_self];
                float* dst = pass == 0 ? gbuf : dx;
#pragma unroll 1
                for (int nt = nt0; nt < NT; nt += 2) {
                    f32x16 g;
#pragma unroll
                    for (int r = 0; r < 16; ++r) g[r] = 0.f;
                    const int k = 32 * nt + i;
                    const bool kok = k < f_in;
                    const float* wp = W + (kok ? k : 0);
                    const float* zp = Zs + (32 * mt + i) * ZS + h;
                    for (int o0 = 0; o0 < 64; o0 += 16) {
                        float a[8], b[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int o = o0 + 2 * u + h;
                            b[u] = wp[(long long)(o < f_out ? o : 0) * f_in];
                            a[u] = zp[o0 + 2 * u];
                        }
#pragma unroll
                        for (int u = 0; u < 8; ++u) {
                            const int o = o0 + 2 * u + h;
                            const float bv = (kok && o < f_out) ? b[u] : 0.f;
                            g = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv, g, 0, 0, 0);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = 32 * mt + mfma32_row(r, lane);
                        if (row < rows && kok) dst[(row0 + row) * f_in + k] = g[r];
                    }
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

// ------------------------------------------------------------------------------------------------ backward, LDS-DMA form
// The fast form of the backward (fp32 rows of 64 or 80 floats, f_out = 64, everything 16-byte aligned).  A workgroup of
// four waves owns 32-atom tiles of ONE degree bucket, as the forward does.  Per step (= tile x gather round) every row
// it needs - the atoms' own rows, NBP neighbour rows each, their dY rows and (ReLU mask) their Y rows - is copied
// HBM -> LDS by global_load_lds_dwordx4 with the forward's chunk swizzle (chunk c of slot s at position c ^ (s & 15)),
// then
//   phase 1  dW[o][k] += sum_r dZ[r][o] * A[r][k]   (A = own rows: W_self, A = neighbour sum: W_rel)
//            the contraction runs over the tile's ATOMS: k-step rp contracts rows rp (lanes 0..31) and rp + 16 (lanes
//            32..63), A operand dZ[row][32 ot + i], B operand A[row][32 kt + i], both plain ds_read_b32 along an LDS
//            row (conflict free: a half-wave reads 8 whole chunks of one row; the swizzle term rp is a compile-time
//            constant: byte offset = (lane part ^ (rp << 4)) + rp * 256).  Wave w owns output tile (ot, kt) =
//            (w & 1, w >> 1); the accumulators stay in registers over all tiles of the workgroup, one atomic flush at
//            the end.  Rows of 80 floats: the 16-column tail [64, 80) is a third k tile whose k-steps the two waves of
//            an `ot` share.
//   phase 2  (NEED_DX) [G | T] = dZ [W_rel | W_self]: wave w owns 32 of the 128 output columns, the weights of the
//            workgroup's degree sit in registers in the permuted k order of the forward (one ds_read_b128 of dZ per
//            four MFMAs); G -> gbuf, T -> dx; the dx kernel below then adds the neighbours' G rows.
// What the measurements on the config-4 batch forced (tools/bench_bwd.py, in-kernel timestamps):
//   * no branch around a DMA or around the load of its index: hipcc puts each in a block of its own and waits for
//     vmcnt(0) in front of the first use of a loaded index there, which serialises the gather (one round trip per
//     plane).  Indices are always loaded for all NBP planes (entries past the list repeat its last one), the DMA
//     bodies are straight-line per plane count (a switch around them).
//   * one MFMA code path: a switch over specialised phase-1 bodies made hipcc keep a second copy of every
//     accumulator.  Neighbour planes that no DMA of the workgroup's degree writes are zeroed once, phase 1 always sums
//     NBP planes; the rows past a bucket's end get their dZ zeroed in LDS.
//   * MFMA operands are prepared one batch AHEAD of the MFMAs that read them, in registers of their own: a VALU write
//     to a register that an MFMA in flight still reads as A/B waits for that MFMA (phase 1: 135 -> 105 cycles/MFMA).
//   * DB: two LDS buffers; the rows of step s + 1 are requested at the top of step s, so a workgroup always has a
//     step's worth of bytes in flight (single buffer: a workgroup alternates between waiting and computing, and the
//     MFMA pipes idle 45 % of the time at two workgroups per CU).
#define DWB 2          // k-steps per operand batch of phase 1

#ifdef JOVA_BWD_PROFILE
__device__ unsigned long long g_bwd_prof[4096 * 8];
extern "C" int jova_debug_bwd_prof(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_bwd_prof), sizeof(unsigned long long) * n);
}
#endif

// DMA requests of one step, straight-line: own rows (XOWN), dY / Y rows (ZLOAD) and CNT neighbour planes.
// Buffer layout (floats): [own | N_0 .. N_{NBP-1} | dY | Y (MASK)] main planes of GT x 64, then (KG > 8) the
// [own | N_0 ..] tail planes of GT x 16.
template <int KG, bool MASK, int NBP, bool XOWN, bool ZLOAD, int CNT>
__device__ __forceinline__ void bwd_issue(const float* __restrict__ x, const float* __restrict__ dy,
                                          const float* __restrict__ y, int n_atoms, int ldx, long long row0, int rows,
                                          const int* nidx, const int* tidx, float* buf, int tid) {
    constexpr int MAINF = GT * 64, TAILF = GT * 16;
    float* const zmain = buf + (1 + NBP) * MAINF;
    float* const ymain = zmain + MAINF;
    float* const xtail = buf + (2 + NBP + (MASK ? 1 : 0)) * MAINF;
    const int wave = tid >> 6, sub = tid >> 4, pos = tid & 15;
    const int chunk = (pos ^ sub) << 2;                  // slot & 15 == sub
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int r = it * 16 + sub;
        const int rc = r < rows ? r : rows - 1;
        const int base = (it * 16 + wave * 4) * 64;
        if (XOWN) glds16(x + (row0 + rc) * ldx + chunk, buf + base);
        if (ZLOAD) {
            glds16(dy + (row0 + rc) * 64 + chunk, zmain + base);
            if (MASK) glds16(y + (row0 + rc) * 64 + chunk, ymain + base);
        }
#pragma unroll
        for (int p = 0; p < CNT; ++p)
            glds16(x + (__umul24(min((unsigned)nidx[2 * p + it], (unsigned)(n_atoms - 1)), (unsigned)ldx) + chunk),
                   buf + (p + 1) * MAINF + base);
    }
    if (KG > 8 && wave < 2) {
        const int r = tid >> 2, tp = tid & 3;
        const int rc = r < rows ? r : rows - 1;
        const int tchunk = 64 + ((tp ^ ((r >> 2) & 3)) << 2);
        const int base = (wave * 16) * 16;
        if (XOWN) glds16(x + (row0 + rc) * ldx + tchunk, xtail + base);
#pragma unroll
        for (int p = 0; p < CNT; ++p)
            glds16(x + (__umul24(min((unsigned)tidx[p], (unsigned)(n_atoms - 1)), (unsigned)ldx) + tchunk),
                   xtail + (p + 1) * TAILF + base);
    }
}

// Raw LDS operands of DWB k-steps (always all NBP neighbour planes: unused planes hold zeros).
template <bool MASK, int NBP>
struct DwBatch {
    float z[DWB], yv[DWB], xs[DWB], sn[DWB][NBP];
    // main image: k-steps rp0 .. rp0 + DWB - 1;  TAIL: k-steps 2 (rp0 + u) + kt of the [slot][16] tail image
    template <bool TAIL>
    __device__ __forceinline__ void load(const char* xp, const char* zmain, const char* ymain, int rp0, int zl,
                                         int xl) {
        constexpr int PL = (TAIL ? GT * 16 : GT * 64) * 4;
#pragma unroll
        for (int u = 0; u < DWB; ++u) {
            // TAIL: kt's share of the row offset and swizzle is folded into zl / xl by the caller
            const int rp = TAIL ? 2 * (rp0 + u) : rp0 + u;
            const int zo = (zl ^ (rp << 4)) + rp * 256;
            const int xo = TAIL ? (xl ^ (((rp >> 2) & 3) << 4)) + rp * 64 : (xl ^ (rp << 4)) + rp * 256;
            z[u] = *reinterpret_cast<const float*>(zmain + zo);
            if (MASK) yv[u] = *reinterpret_cast<const float*>(ymain + zo);
            xs[u] = *reinterpret_cast<const float*>(xp + xo);
#pragma unroll
            for (int p = 0; p < NBP; ++p) sn[u][p] = *reinterpret_cast<const float*>(xp + (p + 1) * PL + xo);
        }
    }
};
// MFMA operands of DWB k-steps, prepared one batch ahead of the MFMAs that read them.
template <bool MASK, int NBP>
struct DwOps {
    float zs[DWB], zz[DWB], xs[DWB], s[DWB];
    __device__ __forceinline__ void prep(const DwBatch<MASK, NBP>& b, bool first) {
#pragma unroll
        for (int u = 0; u < DWB; ++u) {
            float v = b.z[u];
            if (MASK) v = b.yv[u] > 0.f ? v : 0.f;
            zz[u] = v;
            zs[u] = first ? v : 0.f;             // later gather rounds: the own rows / db were done in round 0
            xs[u] = b.xs[u];
            float t = b.sn[u][0];
#pragma unroll
            for (int p = 1; p < NBP; ++p) t += b.sn[u][p];
            s[u] = t;
        }
    }
    template <bool TAIL>
    __device__ __forceinline__ void mma(f32x16& a_self, f32x16& a_rel, float& db_acc) const {
#pragma unroll
        for (int u = 0; u < DWB; ++u) {
            a_self = __builtin_amdgcn_mfma_f32_32x32x2f32(zs[u], xs[u], a_self, 0, 0, 0);
            a_rel = __builtin_amdgcn_mfma_f32_32x32x2f32(zz[u], s[u], a_rel, 0, 0, 0);
            if (!TAIL) db_acc += zs[u];
        }
    }
};

template <int KG, bool MASK, int NBP>
__device__ __forceinline__ void bwd_dw_round(const float* buf, bool first, int i, int h, int ot, int kt,
                                             f32x16& a_self, f32x16& a_rel, f32x16& t_self, f32x16& t_rel,
                                             float& db_acc) {
    constexpr int MAINF = GT * 64;
    const char* xmain = reinterpret_cast<const char*>(buf);
    const char* zmain = reinterpret_cast<const char*>(buf + (1 + NBP) * MAINF);
    const char* ymain = reinterpret_cast<const char*>(buf + (2 + NBP) * MAINF);
    const char* xtail = reinterpret_cast<const char*>(buf + (2 + NBP + (MASK ? 1 : 0)) * MAINF);
    const int e4 = (i & 3) * 4;
    const int zl = h * 4096 + ((8 * ot + (i >> 2)) << 4) + e4;
    const int xl = h * 4096 + ((8 * kt + (i >> 2)) << 4) + e4;
    const int zt = (zl ^ (kt << 4)) + kt * 256;
    const int xt = h * 1024 + (((i >> 2) & 3) << 4) + e4 + kt * 64;
    constexpr int NBM = 16 / DWB;                       // batches of the main image
    constexpr int NBT = KG > 8 ? 8 / DWB : 0;           // batches of the tail image (this wave's share)
    constexpr int NBATCH = NBM + NBT;
    // software pipeline over batches: LDS reads two ahead, operand preparation one ahead of the MFMAs
    DwBatch<MASK, NBP> raw[2];
    DwOps<MASK, NBP> ops[2];
    auto load = [&](DwBatch<MASK, NBP>& r, int b) {
        if (b < NBM) r.template load<false>(xmain, zmain, ymain, b * DWB, zl, xl);
        else r.template load<true>(xtail, zmain, ymain, (b - NBM) * DWB, zt, xt);
    };
    load(raw[0], 0);
    if (NBATCH > 1) load(raw[1], 1);
    ops[0].prep(raw[0], first);
#pragma unroll
    for (int b = 0; b < NBATCH; ++b) {
        if (b + 2 < NBATCH) load(raw[b & 1], b + 2);
        if (b + 1 < NBATCH) ops[(b + 1) & 1].prep(raw[(b + 1) & 1], first);
        __builtin_amdgcn_sched_barrier(0);
        if (b < NBM) ops[b & 1].template mma<false>(a_self, a_rel, db_acc);
        else ops[b & 1].template mma<true>(t_self, t_rel, db_acc);
        __builtin_amdgcn_sched_barrier(0);
    }
}

#define DWP_LD 96          // row stride (floats) of the aligned scratch image of the 21 weight gradients [21][64][96]
__global__ void graphconv_dw_fold_kernel(const float* __restrict__ dwp, GcGrads Gd, int f_in) {
    const int total = 21 * 64 * f_in;
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
        const int k = e % f_in, wo = e / f_in;
        const int o = wo & 63, w = wo >> 6;
        Gd.dw[w][o * f_in + k] += dwp[(long long)wo * DWP_LD + k];
    }
}

template <int KG, bool NEED_DX, bool MASK, int NBP, bool DB>
__global__ void __launch_bounds__(256, 2)
graphconv_bwd_dma_kernel(const float* __restrict__ x, int n_atoms, int ldx, int f_in,
                         const long long* __restrict__ deg_slice, GcParams P, const float* __restrict__ dy,
                         const float* __restrict__ y, GcGrads Gd, float* __restrict__ gbuf, float* __restrict__ dx,
                         int group, BwdPlan plan, float* __restrict__ dwp) {
    constexpr int MAINF = GT * 64, TAILF = GT * 16;
    constexpr int BUFF = (2 + NBP + (MASK ? 1 : 0)) * MAINF + (KG > 8 ? (1 + NBP) * TAILF : 0);
    __shared__ __attribute__((aligned(16))) float lds[(DB ? 2 : 1) * BUFF];
#ifdef JOVA_BWD_PROFILE
    const unsigned long long prof_entry = wall_clock64();
#endif
    int deg, tile0, ntile, bsize;
    long long bstart;
    if (plan.use) {
        if (!locate_tiles_plan(plan, blockIdx.x, deg, bstart, bsize, tile0, ntile)) return;
    } else if (!locate_tiles_batch(deg_slice, blockIdx.x, group, deg, bstart, bsize, tile0, ntile)) {
        return;
    }
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int ot = wave & 1, kt = wave >> 1;
    const int w_rel = deg > 0 ? 2 * (deg - 1) : 20, w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    const int sub = tid >> 4;
    const int* adj_s = deg > 0 ? P.adj[deg - 1] : reinterpret_cast<const int*>(deg_slice);
    const int degm1 = deg > 0 ? deg - 1 : 0;

    int nidx[2 * NBP];
    int tidx[KG > 8 ? NBP : 1];
    auto load_idx = [&](int lrow0, int rows, int nb0) {
#pragma unroll
        for (int p = 0; p < NBP; ++p) {
            const int q = min(nb0 + p, degm1);
#pragma unroll
            for (int it = 0; it < 2; ++it) {
                const int r = it * 16 + sub;
                const int rc = r < rows ? r : rows - 1;
                nidx[2 * p + it] = adj_s[__umul24((unsigned)(lrow0 + rc), (unsigned)deg) + (unsigned)q];
            }
            if (KG > 8) {
                const int r = (tid >> 2) & 31;
                const int rc = r < rows ? r : rows - 1;
                tidx[p] = adj_s[__umul24((unsigned)(lrow0 + rc), (unsigned)deg) + (unsigned)q];
            }
        }
    };
    // step (t, rd) -> buffer b.  Round 0 brings the own rows; dY / Y come with round 0, and with every round when
    // the buffers alternate (DB).
    auto issue = [&](int t, int rd, int b) {
        const int lrow0 = (tile0 + t) * GT;
        const int rows = min(GT, bsize - lrow0);
        const long long row0 = bstart + lrow0;
        const int cnt = min(NBP, deg - rd * NBP);
        float* buf = lds + b * BUFF;
#define JOVA_IS(X_, Z_, C_) bwd_issue<KG, MASK, NBP, X_, Z_, C_>(x, dy, y, n_atoms, ldx, row0, rows, nidx, tidx, buf, tid)
#define JOVA_IS_SW(X_, Z_)                                                                                             \
    switch (cnt) {                                                                                                     \
        case 1: JOVA_IS(X_, Z_, 1); break;                                                                             \
        case 2: JOVA_IS(X_, Z_, (NBP >= 2 ? 2 : NBP)); break;                                                          \
        case 3: JOVA_IS(X_, Z_, (NBP >= 3 ? 3 : NBP)); break;                                                          \
        case 4: JOVA_IS(X_, Z_, (NBP >= 4 ? 4 : NBP)); break;                                                          \
        default: JOVA_IS(X_, Z_, 0); break;                                                                            \
    }
        if (rd == 0) { JOVA_IS_SW(true, true) }
        else if (DB) { JOVA_IS_SW(false, true) }
        else { JOVA_IS_SW(false, false) }
#undef JOVA_IS_SW
#undef JOVA_IS
    };
    // neighbour planes [from, NBP) of buffer b <- 0
    auto zero_planes = [&](int b, int from) {
        float* buf = lds + b * BUFF;
        for (int e = tid; e < (NBP - from) * MAINF; e += 256) buf[(1 + from) * MAINF + e] = 0.f;
        if (KG > 8) {
            float* xtail = buf + (2 + NBP + (MASK ? 1 : 0)) * MAINF;
            for (int e = tid; e < (NBP - from) * TAILF; e += 256) xtail[(1 + from) * TAILF + e] = 0.f;
        }
    };

    // One flat sequence of steps = (tile, gather round); a tile of degree d takes max(1, ceil(d / NBP)) rounds (the
    // product is linear in the neighbour sum).
    const int nrounds = deg > NBP ? (deg + NBP - 1) / NBP : 1;
    const int nsteps = ntile * nrounds;
    const int cnt_last = min(NBP, deg - (nrounds - 1) * NBP);
    if (deg < NBP) {             // planes that no DMA of this workgroup writes hold zeros for its whole life
        zero_planes(0, max(deg, 0));
        if (DB) zero_planes(1, max(deg, 0));
    }
    load_idx(tile0 * GT, min(GT, bsize - tile0 * GT), 0);
    issue(0, 0, 0);
    int t1 = nrounds > 1 ? 0 : 1, rd1 = nrounds > 1 ? 1 : 0;     // step s + 1
    if (nsteps > 1) load_idx((tile0 + t1) * GT, min(GT, bsize - (tile0 + t1) * GT), rd1 * NBP);

    f32x16 a_self, a_rel, t_self, t_rel, g;
#pragma unroll
    for (int r = 0; r < 16; ++r) a_self[r] = a_rel[r] = t_self[r] = t_rel[r] = g[r] = 0.f;
    float db_acc = 0.f;
    // phase 2 weights: lane (i, h) holds W[8 g + 4 h + s][32 cb + i] of W_rel (waves 0, 1) or W_self (waves 2, 3)
    float4 bw[NEED_DX ? 8 : 1];
    const int col2 = 32 * (wave & 1) + i;
    if (NEED_DX) {
        const float* W = P.w[wave < 2 ? w_rel : w_self];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float v[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) v[s] = col2 < f_in ? W[(long long)(8 * q + 4 * h + s) * f_in + col2] : 0.f;
            bw[q] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
#ifdef JOVA_BWD_PROFILE
    const unsigned long long prof_loop = wall_clock64();
    unsigned long long prof_wait = 0, prof_comp = 0;
#endif

    // request step s + 1 (its indices were fetched a step earlier), then fetch the indices of step s + 2
    auto advance = [&](int st) {
        if (st + 1 < nsteps) {
            const int b1 = DB ? ((st + 1) & 1) : 0;
            // a short last round leaves stale neighbour planes in its buffer: zero them (nobody reads that buffer now)
            if (nrounds > 1 && cnt_last < NBP && rd1 == nrounds - 1) zero_planes(b1, cnt_last);
            issue(t1, rd1, b1);
            int t2 = t1, rd2 = rd1 + 1;
            if (rd2 == nrounds) { rd2 = 0; ++t2; }
            if (st + 2 < nsteps) load_idx((tile0 + t2) * GT, min(GT, bsize - (tile0 + t2) * GT), rd2 * NBP);
            t1 = t2;
            rd1 = rd2;
        }
    };

    int t = 0, rd = 0;
    for (int st = 0; st < nsteps; ++st) {
        const int lrow0 = (tile0 + t) * GT;
        const int rows = min(GT, bsize - lrow0);
        const long long row0 = bstart + lrow0;
        float* buf = lds + (DB ? (st & 1) : 0) * BUFF;
        float* zmain = buf + (1 + NBP) * MAINF;
        float* ymain = zmain + MAINF;
        const int tn = t1, rdn = rd1;            // (t, rd) of step s + 1, before advance() moves them on
#ifdef JOVA_BWD_PROFILE
        const unsigned long long p0 = __builtin_readcyclecounter();
#endif
        __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this step's rows landed
        __syncthreads();                         // ... for every wave; every wave is done with the previous step
#ifdef JOVA_BWD_PROFILE
        const unsigned long long p1 = __builtin_readcyclecounter();
#endif
        if (DB) advance(st);
        if (rows < GT) {                         // partial last tile of a bucket: the clamped rows' dZ -> 0
            for (int e = tid; e < (GT - rows) * 64; e += 256) zmain[rows * 64 + e] = 0.f;
            __syncthreads();
        }
        bwd_dw_round<KG, MASK, NBP>(buf, rd == 0, i, h, ot, kt, a_self, a_rel, t_self, t_rel, db_acc);
        if (NEED_DX && rd == 0 && (deg > 0 || wave >= 2)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) g[r] = 0.f;
#pragma unroll
            for (int G = 0; G < 8; ++G) {
                const int off = i * 64 + (((2 * G + h) ^ (i & 15)) << 2);
                float4 a = *reinterpret_cast<const float4*>(zmain + off);
                if (MASK) {
                    const float4 yv = *reinterpret_cast<const float4*>(ymain + off);
                    a.x = yv.x > 0.f ? a.x : 0.f; a.y = yv.y > 0.f ? a.y : 0.f;
                    a.z = yv.z > 0.f ? a.z : 0.f; a.w = yv.w > 0.f ? a.w : 0.f;
                }
                g = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bw[G].x, g, 0, 0, 0);
                g = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bw[G].y, g, 0, 0, 0);
                g = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bw[G].z, g, 0, 0, 0);
                g = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bw[G].w, g, 0, 0, 0);
            }
        }
#ifdef JOVA_BWD_PROFILE
        const unsigned long long p2 = __builtin_readcyclecounter();
        prof_wait += p1 - p0;
        prof_comp += p2 - p1;
#endif
        if (!DB) {
            __syncthreads();                     // this step's LDS reads are done
            advance(st);
        }
        if (NEED_DX && rd == nrounds - 1 && (deg > 0 || wave >= 2)) {
            float* dst = (wave < 2 ? gbuf : dx) + row0 * f_in;        // wave-uniform base, 32-bit lane offsets
            const unsigned fi = (unsigned)f_in, c2 = (unsigned)col2;
            if (col2 < f_in) {
                if (rows == GT) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) dst[(unsigned)mfma32_row(r, lane) * fi + c2] = g[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row = mfma32_row(r, lane);
                        if (row < rows) dst[(unsigned)row * fi + c2] = g[r];
                    }
                }
            }
        }
        t = tn;
        rd = rdn;
    }
#ifdef JOVA_BWD_PROFILE
    const unsigned long long prof_end = wall_clock64();
#endif

    // flush: one atomic per weight element per workgroup
    if (kt == 0) {                               // (every wave summed dZ; the kt = 0 pair covers all 64 columns)
        db_acc += __shfl_xor(db_acc, 32, 64);
        if (lane < 32) {
            atomicAdd(&Gd.db[w_self][32 * ot + i], db_acc);
            if (deg > 0) atomicAdd(&Gd.db[w_rel][32 * ot + i], db_acc);
        }
    }
    // Rows of f_in = 75 floats put every 32-float run of a weight row across two or three 128-byte lines, and the
    // float atomics of ~500 workgroups on such runs took 21 us of the 127 us at 75 -> 64 (13 us with 80-float rows,
    // 2 us with 64-float rows): with a scratch image whose rows are DWP_LD = 96 floats every run is one aligned
    // line; graphconv_dw_fold_kernel adds the image to the real gradients afterwards.
    float* const ws = dwp ? dwp + (long long)w_self * 64 * DWP_LD : Gd.dw[w_self];
    float* const wr = dwp ? dwp + (long long)w_rel * 64 * DWP_LD : Gd.dw[w_rel];
    const int ldw = dwp ? DWP_LD : f_in;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int o = 32 * ot + mfma32_row(r, lane);
        const int k = 32 * kt + i;
        if (k < f_in) {
            atomicAdd(&ws[o * ldw + k], a_self[r]);
            if (deg > 0) atomicAdd(&wr[o * ldw + k], a_rel[r]);
        }
        if (KG > 8 && i < 16 && 64 + i < f_in) {
            atomicAdd(&ws[o * ldw + 64 + i], t_self[r]);
            if (deg > 0) atomicAdd(&wr[o * ldw + 64 + i], t_rel[r]);
        }
    }
#ifdef JOVA_BWD_PROFILE
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (tid == 64 && blockIdx.x < 4096) {
        unsigned long long* o = g_bwd_prof + blockIdx.x * 8;
        o[0] = prof_entry; o[1] = prof_loop; o[2] = prof_end; o[3] = wall_clock64();
        o[4] = prof_wait; o[5] = prof_comp; o[6] = nsteps;
        o[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |
               ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)) << 32);
    }
#endif
}

// dx[i] (holds T_i) += sum of the neighbours' G rows; 16 lanes per atom, float4, four neighbour rows in flight.
// Symmetric adjacency: a gather over i's own list; otherwise the scatter with float atomics.
__global__ void __launch_bounds__(256)
graphconv_bwd_dx_v4_kernel(int n_atoms, int f, const long long* __restrict__ deg_slice, GcParams P,
                           const float* __restrict__ gbuf, float* __restrict__ dx, const int* __restrict__ sym_flag,
                           const int* __restrict__ order);

__global__ void set_int_kernel(int* p, int v) { *p = v; }

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

// ---- 16-lanes-per-row forms (f % 4 == 0): a wave works on 4 atoms at once with float4 loads, the first four
// neighbour rows of each in flight together; used whenever the rows are 16-byte multiples (the model's f = 64).
// Block order (optional, made once per batch by jova_block_order): blockIdx -> 16-row block.  Workgroup b runs on XCD
// b % 8; the order hands each XCD the blocks of one eighth of every degree bucket (by relative position), interleaved
// across buckets, so an atom's row and the reads of it by its neighbours (which sit at about the same relative
// position of their own buckets) meet in ONE L2 close in time instead of in eight L2s a whole bucket sweep apart.
__device__ __forceinline__ int ordered_block(const int* __restrict__ order) {
    return order ? min(max(order[blockIdx.x], 0), (int)gridDim.x - 1) : (int)blockIdx.x;
}

__device__ __forceinline__ void find_bucket(const long long* __restrict__ deg_slice, long long row, int& deg,
                                            long long& lrow) {
    deg = 0;
    lrow = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        const long long s = deg_slice[2 * d], c = deg_slice[2 * d + 1];
        if (row >= s && row < s + c) { deg = d; lrow = row - s; }
    }
}

__global__ void __launch_bounds__(256)
graphpool_fwd_v4_kernel(const float* __restrict__ x, int n_atoms, int f, const long long* __restrict__ deg_slice,
                        GcParams P, float* __restrict__ out, unsigned char* __restrict__ arg,
                        const int* __restrict__ order) {
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)ordered_block(order) * 16 + (threadIdx.x >> 4);
    if (row >= n_atoms) return;
    int deg;
    long long lrow;
    find_bucket(deg_slice, row, deg, lrow);
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    for (int c = 4 * l16; c < f; c += 64) {
        float4 best = *reinterpret_cast<const float4*>(x + row * f + c);
        int ax = 0, ay = 0, az = 0, aw = 0;
        for (int n = 0; n < deg; n += 4) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) j[u] = min(max(arow[min(n + u, deg - 1)], 0), n_atoms - 1);
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (long long)j[u] * f + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (n + u < deg) {
                    if (v[u].x > best.x) { best.x = v[u].x; ax = n + u + 1; }
                    if (v[u].y > best.y) { best.y = v[u].y; ay = n + u + 1; }
                    if (v[u].z > best.z) { best.z = v[u].z; az = n + u + 1; }
                    if (v[u].w > best.w) { best.w = v[u].w; aw = n + u + 1; }
                }
            }
        }
        *reinterpret_cast<float4*>(out + row * f + c) = best;
        *reinterpret_cast<unsigned*>(arg + row * f + c) = (unsigned)ax | ((unsigned)ay << 8) | ((unsigned)az << 16) |
                                                          ((unsigned)aw << 24);
    }
}

// symmetric adjacency only (the host wrapper falls back to the generic kernel otherwise)
__global__ void __launch_bounds__(256)
graphpool_bwd_v4_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int n_atoms, int f,
                        const long long* __restrict__ deg_slice, GcParams P, float* __restrict__ dx) {
    const int l16 = threadIdx.x & 15;
    const int grp = (threadIdx.x & 63) >> 4;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool rok = row < n_atoms;
    int deg = 0;
    long long lrow = 0;
    if (rok) find_bucket(deg_slice, row, deg, lrow);
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    // position of this atom inside each neighbour's own list: lane m of the group probes entry m, a ballot finds it
    int nbr[10];
    unsigned pos[10];
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        int j = -1, hit = 0;
        if (n < deg) {
            j = min(max(arow[n], 0), n_atoms - 1);
            int dj;
            long long lj;
            find_bucket(deg_slice, j, dj, lj);
            if (l16 < dj) hit = P.adj[dj - 1][lj * dj + l16] == (int)row;
        }
        const unsigned long long b = __ballot(hit);
        const unsigned m16 = (unsigned)(b >> (16 * grp)) & 0xFFFFu;
        nbr[n] = j;
        pos[n] = m16 ? (unsigned)__ffs(m16) : 0xFFu;       // 1-based position = the arg-max code of that neighbour
    }
    if (!rok) return;
    for (int c = 4 * l16; c < f; c += 64) {
        const unsigned a0 = *reinterpret_cast<const unsigned*>(arg + row * f + c);
        const float4 g0 = *reinterpret_cast<const float4*>(dy + row * f + c);
        float4 s;
        s.x = (a0 & 0xFF) == 0 ? g0.x : 0.f;
        s.y = ((a0 >> 8) & 0xFF) == 0 ? g0.y : 0.f;
        s.z = ((a0 >> 16) & 0xFF) == 0 ? g0.z : 0.f;
        s.w = (a0 >> 24) == 0 ? g0.w : 0.f;
#pragma unroll
        for (int n = 0; n < 10; ++n) {
            if (n < deg) {
                const long long j = nbr[n];
                const unsigned a = *reinterpret_cast<const unsigned*>(arg + j * f + c);
                const float4 g = *reinterpret_cast<const float4*>(dy + j * f + c);
                const unsigned p = pos[n];
                s.x += (a & 0xFF) == p ? g.x : 0.f;
                s.y += ((a >> 8) & 0xFF) == p ? g.y : 0.f;
                s.z += ((a >> 16) & 0xFF) == p ? g.z : 0.f;
                s.w += (a >> 24) == p ? g.w : 0.f;
            }
        }
        *reinterpret_cast<float4*>(dx + row * f + c) = s;
    }
}

// bucket of `row` plus the offset of its adjacency row inside the concatenation adj_1 | adj_2 | .. | adj_10
__device__ __forceinline__ void find_bucket_off(const long long* __restrict__ deg_slice, long long row, int& deg,
                                                long long& lrow, long long& eoff) {
    deg = 0;
    lrow = 0;
    eoff = 0;
    long long run = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        const long long s = deg_slice[2 * d], c = deg_slice[2 * d + 1];
        if (row >= s && row < s + c) { deg = d; lrow = row - s; eoff = run + (row - s) * d; }
        run += c * d;
    }
}

// rev_pos[e] for edge e = (atom i, its n-th neighbour j): the 1-based position of i inside j's own list (0xFF: not
// there).  A property of the batch graph, built once per batch; with it the backward of GraphPool is a plain gather.
__global__ void __launch_bounds__(256)
adj_reverse_pos_kernel(int n_atoms, const long long* __restrict__ deg_slice, GcParams P,
                       unsigned char* __restrict__ rev_pos) {
    const int l16 = threadIdx.x & 15;
    const int grp = (threadIdx.x & 63) >> 4;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool rok = row < n_atoms;
    int deg = 0;
    long long lrow = 0, eoff = 0;
    if (rok) find_bucket_off(deg_slice, row, deg, lrow, eoff);
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    unsigned mine = 0xFFu;
#pragma unroll
    for (int n = 0; n < 10; ++n) {
        int hit = 0;
        if (n < deg) {
            const int j = min(max(arow[n], 0), n_atoms - 1);
            int dj;
            long long lj;
            find_bucket(deg_slice, j, dj, lj);
            if (l16 < dj) hit = P.adj[dj - 1][lj * dj + l16] == (int)row;
        }
        const unsigned long long b = __ballot(hit);
        const unsigned m16 = (unsigned)(b >> (16 * grp)) & 0xFFFFu;
        if (l16 == n) mine = m16 ? (unsigned)__ffs(m16) : 0xFFu;
    }
    if (rok && l16 < deg) rev_pos[eoff + l16] = (unsigned char)mine;
}

// GraphPool backward as a gather with the precomputed reverse positions: 16 lanes per atom, float4, the (arg, dy)
// rows of four neighbours in flight together.
__global__ void __launch_bounds__(256)
graphpool_bwd_v5_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int n_atoms, int f,
                        const long long* __restrict__ deg_slice, GcParams P,
                        const unsigned char* __restrict__ rev_pos, float* __restrict__ dx,
                        const int* __restrict__ order) {
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)ordered_block(order) * 16 + (threadIdx.x >> 4);
    if (row >= n_atoms) return;
    int deg;
    long long lrow, eoff;
    find_bucket_off(deg_slice, row, deg, lrow, eoff);
    const int* arow = deg > 0 ? P.adj[deg - 1] + lrow * deg : nullptr;
    const unsigned char* prow = rev_pos + eoff;
    for (int c = 4 * l16; c < f; c += 64) {
        const unsigned a0 = *reinterpret_cast<const unsigned*>(arg + row * f + c);
        const float4 g0 = *reinterpret_cast<const float4*>(dy + row * f + c);
        float4 s;
        s.x = (a0 & 0xFF) == 0 ? g0.x : 0.f;
        s.y = ((a0 >> 8) & 0xFF) == 0 ? g0.y : 0.f;
        s.z = ((a0 >> 16) & 0xFF) == 0 ? g0.z : 0.f;
        s.w = (a0 >> 24) == 0 ? g0.w : 0.f;
        for (int n = 0; n < deg; n += 4) {
            long long j[4];
            unsigned p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nn = min(n + u, deg - 1);
                j[u] = min(max(arow[nn], 0), n_atoms - 1);
                p[u] = n + u < deg ? (unsigned)prow[nn] : 0x100u;      // 0x100 matches no arg-max code
            }
            unsigned a[4];
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const unsigned*>(arg + j[u] * f + c);
                g[u] = *reinterpret_cast<const float4*>(dy + j[u] * f + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s.x += (a[u] & 0xFF) == p[u] ? g[u].x : 0.f;
                s.y += ((a[u] >> 8) & 0xFF) == p[u] ? g[u].y : 0.f;
                s.z += ((a[u] >> 16) & 0xFF) == p[u] ? g[u].z : 0.f;
                s.w += (a[u] >> 24) == p[u] ? g[u].w : 0.f;
            }
        }
        *reinterpret_cast<float4*>(dx + row * f + c) = s;
    }
}

// Row table (optional, made once per batch by jova_row_table): slot = launch block * 16 + row-in-block holds
//   { row | -1, degree, the first four neighbours, their four reverse positions packed one per byte, 0 }
// so a 16-lane group starts its gathers after ONE 32-byte read instead of after the chain block order -> degree
// buckets -> adjacency pointer -> adjacency row (four dependent round trips that left the pool kernels latency-bound).
// Neighbours five and up (rare: degree > 4) are read from the adjacency lists as before.
__global__ void __launch_bounds__(256)
row_table_kernel(int n_atoms, int n_slots, const long long* __restrict__ deg_slice, GcParams P,
                 const unsigned char* __restrict__ rev_pos, const int* __restrict__ order, int n_blocks,
                 int4* __restrict__ table) {
    const int slot = blockIdx.x * 256 + threadIdx.x;
    if (slot >= n_slots) return;
    const int b = slot >> 4;
    const int blk = order ? min(max(order[b], 0), n_blocks - 1) : b;
    const long long row = (long long)blk * 16 + (slot & 15);
    int4 t0 = make_int4(-1, 0, 0, 0), t1 = make_int4(0, 0, (int)0xFFFFFFFFu, 0);
    if (row < n_atoms) {
        int deg;
        long long lrow, eoff;
        find_bucket_off(deg_slice, row, deg, lrow, eoff);
        int j[4] = {0, 0, 0, 0};
        unsigned pk = 0xFFFFFFFFu;
        if (deg > 0) {
            const int* arow = P.adj[deg - 1] + lrow * deg;
            pk = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int nn = min(u, deg - 1);
                j[u] = min(max(arow[nn], 0), n_atoms - 1);
                const unsigned p = (u < deg && rev_pos) ? (unsigned)rev_pos[eoff + nn] : 0xFFu;
                pk |= p << (8 * u);
            }
        }
        t0 = make_int4((int)row, deg, j[0], j[1]);
        t1 = make_int4(j[2], j[3], (int)pk, 0);
    }
    table[2 * (long long)slot] = t0;
    table[2 * (long long)slot + 1] = t1;
}

__global__ void __launch_bounds__(256)
graphpool_fwd_t_kernel(const float* __restrict__ x, int n_atoms, int f, const long long* __restrict__ deg_slice,
                       GcParams P, float* __restrict__ out, unsigned char* __restrict__ arg,
                       const int4* __restrict__ table) {
    const int l16 = threadIdx.x & 15;
    const long long slot = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int4 t0 = table[2 * slot], t1 = table[2 * slot + 1];
    const long long row = t0.x;
    if (row < 0 || row >= n_atoms) return;
    const int deg = min(max(t0.y, 0), 10);
    const int j0[4] = {min(max(t0.z, 0), n_atoms - 1), min(max(t0.w, 0), n_atoms - 1), min(max(t1.x, 0), n_atoms - 1),
                       min(max(t1.y, 0), n_atoms - 1)};
    const int* arow = nullptr;
    if (deg > 4) {
        int d2;
        long long lrow;
        find_bucket(deg_slice, row, d2, lrow);
        arow = P.adj[deg - 1] + lrow * deg;
    }
    for (int c = 4 * l16; c < f; c += 64) {
        float4 best = *reinterpret_cast<const float4*>(x + row * f + c);
        int ax = 0, ay = 0, az = 0, aw = 0;
        for (int n = 0; n < deg; n += 4) {
            int j[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                j[u] = n == 0 ? j0[u] : min(max(arow[min(n + u, deg - 1)], 0), n_atoms - 1);
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(x + (long long)j[u] * f + c);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (n + u < deg) {
                    if (v[u].x > best.x) { best.x = v[u].x; ax = n + u + 1; }
                    if (v[u].y > best.y) { best.y = v[u].y; ay = n + u + 1; }
                    if (v[u].z > best.z) { best.z = v[u].z; az = n + u + 1; }
                    if (v[u].w > best.w) { best.w = v[u].w; aw = n + u + 1; }
                }
            }
        }
        *reinterpret_cast<float4*>(out + row * f + c) = best;
        *reinterpret_cast<unsigned*>(arg + row * f + c) = (unsigned)ax | ((unsigned)ay << 8) | ((unsigned)az << 16) |
                                                          ((unsigned)aw << 24);
    }
}

__global__ void __launch_bounds__(256)
graphpool_bwd_t_kernel(const float* __restrict__ dy, const unsigned char* __restrict__ arg, int n_atoms, int f,
                       const long long* __restrict__ deg_slice, GcParams P,
                       const unsigned char* __restrict__ rev_pos, float* __restrict__ dx,
                       const int4* __restrict__ table) {
    const int l16 = threadIdx.x & 15;
    const long long slot = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    const int4 t0 = table[2 * slot], t1 = table[2 * slot + 1];
    const long long row = t0.x;
    if (row < 0 || row >= n_atoms) return;
    const int deg = min(max(t0.y, 0), 10);
    const long long j0[4] = {min(max(t0.z, 0), n_atoms - 1), min(max(t0.w, 0), n_atoms - 1),
                             min(max(t1.x, 0), n_atoms - 1), min(max(t1.y, 0), n_atoms - 1)};
    const unsigned pk = (unsigned)t1.z;
    const int* arow = nullptr;
    const unsigned char* prow = nullptr;
    if (deg > 4) {
        int d2;
        long long lrow, eoff;
        find_bucket_off(deg_slice, row, d2, lrow, eoff);
        arow = P.adj[deg - 1] + lrow * deg;
        prow = rev_pos + eoff;
    }
    for (int c = 4 * l16; c < f; c += 64) {
        const unsigned a0 = *reinterpret_cast<const unsigned*>(arg + row * f + c);
        const float4 g0 = *reinterpret_cast<const float4*>(dy + row * f + c);
        float4 s;
        s.x = (a0 & 0xFF) == 0 ? g0.x : 0.f;
        s.y = ((a0 >> 8) & 0xFF) == 0 ? g0.y : 0.f;
        s.z = ((a0 >> 16) & 0xFF) == 0 ? g0.z : 0.f;
        s.w = (a0 >> 24) == 0 ? g0.w : 0.f;
        for (int n = 0; n < deg; n += 4) {
            long long j[4];
            unsigned p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (n == 0) {
                    j[u] = j0[u];
                    p[u] = u < deg ? (pk >> (8 * u)) & 0xFFu : 0x100u;      // 0x100 matches no arg-max code
                } else {
                    const int nn = min(n + u, deg - 1);
                    j[u] = min(max(arow[nn], 0), n_atoms - 1);
                    p[u] = n + u < deg ? (unsigned)prow[nn] : 0x100u;
                }
            }
            unsigned a[4];
            float4 g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = *reinterpret_cast<const unsigned*>(arg + j[u] * f + c);
                g[u] = *reinterpret_cast<const float4*>(dy + j[u] * f + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                s.x += (a[u] & 0xFF) == p[u] ? g[u].x : 0.f;
                s.y += ((a[u] >> 8) & 0xFF) == p[u] ? g[u].y : 0.f;
                s.z += ((a[u] >> 16) & 0xFF) == p[u] ? g[u].z : 0.f;
                s.w += (a[u] >> 24) == p[u] ? g[u].w : 0.f;
            }
        }
        *reinterpret_cast<float4*>(dx + row * f + c) = s;
    }
}

__global__ void __launch_bounds__(256)
graphconv_bwd_dx_v4_kernel(int n_atoms, int f, const long long* __restrict__ deg_slice, GcParams P,
                           const float* __restrict__ gbuf, float* __restrict__ dx, const int* __restrict__ sym_flag,
                           const int* __restrict__ order) {
    const int l16 = threadIdx.x & 15;
    const long long row = (long long)ordered_block(order) * 16 + (threadIdx.x >> 4);
    if (row >= n_atoms) return;
    int deg;
    long long lrow;
    find_bucket(deg_slice, row, deg, lrow);
    if (deg == 0) return;
    const int* arow = P.adj[deg - 1] + lrow * deg;
    if (*sym_flag != 0) {
        for (int c = 4 * l16; c < f; c += 64) {
            float4 acc = *reinterpret_cast<const float4*>(dx + row * f + c);
            for (int n = 0; n < deg; n += 4) {
                int j[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) j[u] = min(max(arow[min(n + u, deg - 1)], 0), n_atoms - 1);
                float4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(gbuf + (long long)j[u] * f + c);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (n + u < deg) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
                }
            }
            *reinterpret_cast<float4*>(dx + row * f + c) = acc;
        }
    } else {
        for (int c = 4 * l16; c < f; c += 64) {
            const float4 g = *reinterpret_cast<const float4*>(gbuf + row * f + c);
            for (int n = 0; n < deg; ++n) {
                float* d = dx + (long long)min(max(arow[n], 0), n_atoms - 1) * f + c;
                atomicAdd(d, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
            }
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

// float4 forms (g % 4 == 0, 16-byte aligned): one thread per 16 bytes of the padded tensor, one division per 16 bytes
__global__ void __launch_bounds__(256)
gather2d_fwd_v4_kernel(const float* __restrict__ x, const int* __restrict__ mol_off, int a_max, int bsz, int g4,
                       float* __restrict__ out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;       // float4 index
    const long long total = (long long)a_max * bsz * g4;
    if (e >= total) return;
    const unsigned ab = (unsigned)(e / g4);
    const int c = (int)(e - (long long)ab * g4);
    const int a = (int)(ab / (unsigned)bsz), b = (int)(ab - (unsigned)a * (unsigned)bsz);
    const int o = mol_off[b], n = mol_off[b + 1] - o;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a < n) v = reinterpret_cast<const float4*>(x)[(long long)(o + a) * g4 + c];
    reinterpret_cast<float4*>(out)[e] = v;
}

__global__ void __launch_bounds__(256)
gather2d_bwd_v4_kernel(const float* __restrict__ dout, const int* __restrict__ mol_off, int a_max, int bsz, int g4,
                       float* __restrict__ dx) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a_max * bsz * g4;
    if (e >= total) return;
    const unsigned ab = (unsigned)(e / g4);
    const int c = (int)(e - (long long)ab * g4);
    const int a = (int)(ab / (unsigned)bsz), b = (int)(ab - (unsigned)a * (unsigned)bsz);
    const int o = mol_off[b], n = mol_off[b + 1] - o;
    if (a < n) reinterpret_cast<float4*>(dx)[(long long)(o + a) * g4 + c] = reinterpret_cast<const float4*>(dout)[e];
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
#define NS_RESIDENT 4
// VEC = 4: every access is 16 bytes (n % 4 == 0, 16-byte aligned buffers); element order inside the sums differs from
// VEC = 1 only in which lane adds which element - the per-workgroup partials are fp64 and combined in a fixed order.
// RES > 0 (VEC = 4): every thread owns at most RES 16-byte pieces, which then stay in registers from the first evaluation
// to the last (x, the iterate and its predecessor): nothing but the barrier words moves between evaluations.  Same
// elements per thread, same order of the sums as the memory form, so the results are the same bits.
template <int VEC, int RES>
__global__ void __launch_bounds__(NS_THREADS)
nonsat_persistent_kernel(const float* __restrict__ x, float* __restrict__ ya, float* __restrict__ yb, long long n,
                         float ep, int max_iter, double* partial, int* ctrl, float* __restrict__ out,
                         float* __restrict__ dfac, int* __restrict__ n_eval_out, unsigned long long spin_limit) {
    __shared__ double red[NS_THREADS / 64];
    __shared__ int s_flag;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int nb = gridDim.x;
    const long long stride = (long long)nb * NS_THREADS * VEC;
    const long long e0 = ((long long)blockIdx.x * NS_THREADS + tid) * VEC;
    int evals = 0;
    bool failed = false;
    constexpr int NR = RES > 0 ? RES : 1;
    float xr[NR][VEC], yc[NR][VEC], yv[NR][VEC];          // resident form: x, current iterate, its predecessor
    if (RES > 0) {
#pragma unroll
        for (int sl = 0; sl < NR; ++sl) {
            const long long e = e0 + sl * stride;
            if (e < n) *reinterpret_cast<float4*>(xr[sl]) = *reinterpret_cast<const float4*>(x + e);
#pragma unroll
            for (int u = 0; u < VEC; ++u) yc[sl][u] = yv[sl][u] = xr[sl][u];
        }
    }
    for (int it = 0;; ++it) {
        const float* yin = (it & 1) ? yb : ya;
        float* yout = (it & 1) ? ya : yb;
        double acc = 0.0;
        if (RES > 0) {
#pragma unroll
            for (int sl = 0; sl < NR; ++sl) {
                if (e0 + sl * stride < n) {
#pragma unroll
                    for (int u = 0; u < VEC; ++u) {
                        const float y = yc[sl][u];
                        const float yn = (2.f * (y * y * y) / 3.f + xr[sl][u]) / (y * y + 1.f);
                        acc += (double)fabsf(yn - y);
                        yv[sl][u] = y;
                        yc[sl][u] = yn;
                    }
                }
            }
        }
        for (long long e = e0; RES == 0 && e < n; e += stride) {
            float xv[VEC], y[VEC], yn[VEC];
            if (VEC == 4) {
                *reinterpret_cast<float4*>(xv) = *reinterpret_cast<const float4*>(x + e);
                if (it == 0) *reinterpret_cast<float4*>(y) = *reinterpret_cast<const float4*>(xv);
                else *reinterpret_cast<float4*>(y) = *reinterpret_cast<const float4*>(yin + e);
            } else {
                xv[0] = x[e];
                y[0] = it == 0 ? xv[0] : yin[e];
            }
#pragma unroll
            for (int u = 0; u < VEC; ++u) {
                yn[u] = (2.f * (y[u] * y[u] * y[u]) / 3.f + xv[u]) / (y[u] * y[u] + 1.f);
                acc += (double)fabsf(yn[u] - y[u]);       // every |delta| enters the fp64 sum on its own, as before
            }
            if (VEC == 4) *reinterpret_cast<float4*>(yout + e) = *reinterpret_cast<const float4*>(yn);
            else yout[e] = yn[0];
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
            // The give-up is COLLECTIVE: whoever runs out of patience raises ctrl[2]; every spinner polls it, and everyone
            // looks at it once more after the match, so no workgroup iterates on alone.  A workgroup that leaves this way
            // poisons its share of the outputs (NaN) below: a torn activation cannot flow into training unnoticed.
            int ok = 1;
            while (__hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                if (__hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { ok = 0; break; }
                __builtin_amdgcn_s_sleep(2);
                if (wall_clock64() - t0 > spin_limit) {                        // default 2 s at 100 MHz
                    __hip_atomic_store(&ctrl[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            if (ok && __hip_atomic_load(&ctrl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) ok = 0;
            s_flag = ok;
        }
        __syncthreads();
        if (!s_flag) {
            failed = true;
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
    if (tid == 0) {
        if (blockIdx.x == 0) {
            n_eval_out[0] = evals;
            n_eval_out[1] = __hip_atomic_load(&ctrl[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (failed) n_eval_out[2] = 1;           // (zeroed by the host wrapper; only ever raised)
    }
    if (failed) {
        const float qnan = __builtin_nanf("");
        for (long long e = e0; e < n; e += stride)
            for (int u = 0; u < VEC; ++u) { out[e + u] = qnan; dfac[e + u] = qnan; }
        return;
    }
    if (RES > 0) {
#pragma unroll
        for (int sl = 0; sl < NR; ++sl) {
            const long long e = e0 + sl * stride;
            if (e < n) {
                *reinterpret_cast<float4*>(out + e) = *reinterpret_cast<const float4*>(yc[sl]);
                *reinterpret_cast<float4*>(dfac + e) =
                    make_float4(1.f / (yv[sl][0] * yv[sl][0] + 1.f), 1.f / (yv[sl][1] * yv[sl][1] + 1.f),
                                1.f / (yv[sl][VEC > 1 ? 2 : 0] * yv[sl][VEC > 1 ? 2 : 0] + 1.f),
                                1.f / (yv[sl][VEC > 1 ? 3 : 0] * yv[sl][VEC > 1 ? 3 : 0] + 1.f));
            }
        }
        return;
    }
    // evaluation evals-1 wrote (evals-1 even ? yb : ya); its input y_prev is x (evals == 1) or the other buffer
    const bool res_in_b = ((evals - 1) & 1) == 0;
    const float* yr = res_in_b ? yb : ya;
    const float* yp = evals == 1 ? x : (res_in_b ? ya : yb);
    for (long long e = e0; e < n; e += stride) {
        // d y_final / d x = 1 / (y_prev^2 + 1): the gradient flows through the last evaluation only
        if (VEC == 4) {
            const float4 pv = *reinterpret_cast<const float4*>(yp + e);
            *reinterpret_cast<float4*>(out + e) = *reinterpret_cast<const float4*>(yr + e);
            *reinterpret_cast<float4*>(dfac + e) = make_float4(1.f / (pv.x * pv.x + 1.f), 1.f / (pv.y * pv.y + 1.f),
                                                               1.f / (pv.z * pv.z + 1.f), 1.f / (pv.w * pv.w + 1.f));
        } else {
            const float pv = yp[e];
            out[e] = yr[e];
            dfac[e] = 1.f / (pv * pv + 1.f);
        }
    }
}

__global__ void mul_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                           float* __restrict__ out) {
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x)
        out[e] = a[e] * b[e];
}

// ------------------------------------------------------------------------------------------------ forward, wave-private form
// Second-generation forward (fp32 rows of 64 or 80 floats, f_out = 64): no workgroup barrier in the loop at all.
//   * a WAVE owns 16-atom tiles (v_mfma_f32_16x16x4_f32: 16 atoms x 16 output features per instruction, 32 cycles,
//     four independent accumulators = the 64 output features), eight waves per workgroup, one workgroup per CU, all of
//     one degree bucket; the bucket's two weight matrices sit in LDS once per workgroup (fragment order, one
//     conflict-free ds_read_b128 per four MFMAs), so a wave needs ~150 registers and two waves share a SIMD: while one
//     waits for its rows the other owns the matrix pipe.
//   * rows travel HBM/L2 -> the wave's own LDS planes by global_load_lds_dwordx4 (inline asm: completion is tracked by
//     hand with s_waitcnt vmcnt(N), nothing else in the workgroup is involved), chunk c of row r at position c ^ r, and
//     are read back as MFMA fragments straight into registers (own rows X, neighbour sum S formed while reading).
//     Once a tile's fragments are in registers its planes are free: the DMA of the wave's NEXT tile, the index loads of
//     the tile after that and the four stores of the PREVIOUS tile are issued one by one between the groups of MFMAs of
//     the current tile (128 MFMAs = 4 096 cycles), so the vector-memory issue stalls (100-200 cycles per instruction,
//     measured) sit in the shadow of the matrix pipe instead of in front of it.
//   * out^T = W x^T: the weights are the A operand and the atoms the B operand, so a lane ends up with four CONSECUTIVE
//     output features of one atom: one 16-byte store per accumulator, 64 contiguous bytes per atom and instruction.
//   * k order permuted as in the 32-atom form: MFMA step (m, s) contracts features 16 m + 4 q + s (q = lane >> 4).
//   * work split on the host (WavePlan): workgroups per bucket in proportion to its cost, at most one workgroup per CU;
//     inside a bucket workgroup b takes the chunk whose relative position matches its XCD (b % 8), so the neighbour rows
//     (same molecule, about the same relative position in their own buckets) are fetched into ONE L2.
#define WT 16
struct WavePlan {
    int first[NB + 1];       // workgroups [first[k], first[k + 1]) work on bucket of degree 10 - k
    int start[NB];           // by degree
    int size[NB];
};

__device__ __forceinline__ void dma16_m0(const float* g, unsigned lds) {        // lds: wave-uniform LDS byte address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
// the same with a wave-uniform base pointer (SGPR pair) and a 32-bit per-lane byte offset: no 64-bit address registers
__device__ __forceinline__ void dma16_m0_s(const float* base, unsigned off, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off), "s"(base), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_byte_addr(const float* p) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)p);
}
typedef float f32x4v __attribute__((ext_vector_type(4)));

// The 16 NM (x 2 with neighbours) MFMAs of one 16-atom tile.  Weight fragments of step m + 1 are read from LDS while
// the MFMAs of step m run; every accumulator is reused after four independent MFMAs (40-cycle latency).  `side(k)` is
// called in front of every group of 4 (8) MFMAs, k = 0 .. 4 NM - 1: the wave's memory traffic for its OTHER tiles (DMA
// of the next tile's rows, stores of the previous tile's results) is issued there, in the shadow of the matrix pipe.
template <int NM, bool REL, typename Side>
__device__ __forceinline__ void wave_mfma_tile(const float* wl0, const float* wl1, const f32x4v (&xf)[NM],
                                               const f32x4v (&sf)[NM], f32x4v (&acc)[4], Side&& side) {
    f32x4v ws[2][4], wr[2][4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
        ws[0][fb] = *reinterpret_cast<const f32x4v*>(wl0 + (fb * NM) * 256);
        if (REL) wr[0][fb] = *reinterpret_cast<const f32x4v*>(wl1 + (fb * NM) * 256);
    }
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int c = m & 1, nx = c ^ 1;
        if (m + 1 < NM) {
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) {
                ws[nx][fb] = *reinterpret_cast<const f32x4v*>(wl0 + (fb * NM + m + 1) * 256);
                if (REL) wr[nx][fb] = *reinterpret_cast<const f32x4v*>(wl1 + (fb * NM + m + 1) * 256);
            }
        }
        __builtin_amdgcn_sched_barrier(0);           // reads of step m + 1 are issued in front of the MFMAs of step m
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            side(4 * m + s);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int fb = 0; fb < 4; ++fb)
                acc[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[c][fb][s], xf[m][s], acc[fb], 0, 0, 0);
            if (REL) {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
                    acc[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wr[c][fb][s], sf[m][s], acc[fb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifndef JOVA_WAVE_STAGGER
#define JOVA_WAVE_STAGGER 64          // x 64 cycles
#endif
#ifdef JOVA_WAVE_PROFILE
__device__ unsigned long long g_wave_prof[256 * 8 * 8 + 512];
extern "C" int jova_debug_wave_prof(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_prof), sizeof(unsigned long long) * n);
}
#define WPROF(var) const unsigned long long var = __builtin_readcyclecounter()
#else
#define WPROF(var)
#endif

template <int KG, int NPL, bool STATS>
__global__ void __launch_bounds__(512, 1)
graphconv_fwd_wave_kernel(const float* __restrict__ x, int n_atoms, int ldx, int f_in, GcParams P, int relu, float* __restrict__ out, double* __restrict__ stats, WavePlan plan) {
    constexpr int NM = KG / 2;                       // b128 fragment reads per row: 16 features each
    constexpr int NS = 4 * NM;                       // side slots of a tile's MFMA phase
    constexpr int NBR = NPL - 1;                     // neighbour planes per gather round
    constexpr int PPP = KG > 8 ? 5 : 4;              // DMA pieces (1 KiB wave-instructions) per plane
    constexpr int TAIL_F = KG > 8 ? WT * 16 : 0;
    constexpr int PLANE_F = WT * 64 + TAIL_F;        // floats per plane: [16 rows][64] main image (+ [16][16] tail image)
    constexpr int WMAT_F = 4 * NM * 256;             // floats per weight matrix image
    static_assert(NPL * PPP <= NS, "every DMA piece of a tile needs a side slot");
    __shared__ __attribute__((aligned(16))) float lds[2 * WMAT_F + 8 * NPL * PLANE_F];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
#ifdef JOVA_WAVE_PROFILE
    const unsigned long long pf_entry = __builtin_readcyclecounter();
    const unsigned long long pf_wall0 = wall_clock64();
#endif

    // ---- this workgroup's bucket and its chunk of the bucket's tiles
    int kb = 0;
#pragma unroll
    for (int k = 1; k < NB; ++k) kb += ((int)blockIdx.x >= plan.first[k]) ? 1 : 0;
    const int deg = NB - 1 - kb;
    const int wg0 = plan.first[kb], nwg = plan.first[kb + 1] - wg0;
    int bstart = 0, bsize = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        if (d == deg) { bstart = plan.start[d]; bsize = plan.size[d]; }
    }
    // XCD x (= blockIdx % 8) owns the x-th share of the bucket's tiles (shares in proportion to the XCD's workgroups of
    // this bucket); inside the share the tiles are dealt round-robin to ALL its waves, so at any moment the XCD's waves
    // of every bucket work at about the same relative position of their buckets: the rows one bucket streams are the
    // neighbour rows another gathers, and they meet in that XCD's L2 (a contiguous chunk per workgroup let the buckets
    // drift apart in time; measured: profiles/r02_pmc_graph.json).
    int before = 0, mine = 0, rank = 0;
    {
        const int xcd = blockIdx.x & 7, j = (int)blockIdx.x - wg0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j0 = (k - wg0) & 7;
            const int cnt = j0 < nwg ? (nwg - j0 + 7) >> 3 : 0;
            before += k < xcd ? cnt : 0;
            mine = k == xcd ? cnt : mine;
        }
        rank = (j - ((xcd - wg0) & 7)) >> 3;
    }
    const int tiles = (bsize + WT - 1) / WT;
    const int t_begin = (int)((long long)tiles * before / nwg) + rank * 8;
    const int t_end = (int)((long long)tiles * (before + mine) / nwg);
    const int tstep = mine * 8;
    // tiles of this wave: t_begin + wave + tstep n  (claiming them from a per-workgroup counter was measured: the atomic's
    // round trip costs more than the <= 1 tile of imbalance it removes)
    const int t0 = t_begin + wave;

    float* const pl = lds + 2 * WMAT_F + wave * (NPL * PLANE_F);
    const unsigned pl_addr = lds_byte_addr(pl);
    const int* adj_d = deg > 0 ? P.adj[deg - 1] : nullptr;
    const unsigned nmax = (unsigned)(n_atoms - 1);
    const int swzt = ((4 - (i >> 2)) & 3);           // tail image swizzle by row group: 0, 3, 2, 1
    const int cnt0 = min(deg, NBR);                  // neighbour planes of a tile's first gather round

    // neighbour row numbers of a gather round, in the DMA's lane -> row map (main image: lane L covers row 4 j + (L >> 4)
    // with piece j; tail image: row L >> 2).  Only the planes the round uses are loaded.
    int ix[NBR][4];
    int ixt[KG > 8 ? NBR : 1];
    auto load_idx = [&](int t, int nb0) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
#pragma unroll
        for (int p = 0; p < NBR; ++p) {
            if (nb0 + p < deg) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = min(4 * j + q, rows - 1);
                    ix[p][j] = adj_d[(lrow0 + r) * deg + nb0 + p];
                }
                if (KG > 8) {
                    const int r = min(lane >> 2, rows - 1);
                    ixt[p] = adj_d[(lrow0 + r) * deg + nb0 + p];
                }
            }
        }
    };
    // DMA piece j of a plane (j < 4: rows 4 j .. 4 j + 3 of the main image, j == 4: the tail image)
    auto issue_piece = [&](int pp, int j, int row) {
        if (j < 4) {
            const unsigned r = 4 * j + q;
            const float* g = x + ((size_t)min((unsigned)row, nmax) * (unsigned)ldx + (((lane & 15) ^ r) << 2));
            dma16_m0(g, pl_addr + (pp * PLANE_F + j * 256) * 4);
        } else {
            const int sw = (4 - (lane >> 4)) & 3;                     // row group of row (lane >> 2)
            const float* g = x + ((size_t)min((unsigned)row, nmax) * (unsigned)ldx + 64 + (((lane & 3) ^ sw) << 2));
            dma16_m0(g, pl_addr + (pp * PLANE_F + WT * 64) * 4);
        }
    };
    // piece k (= plane k / PPP, piece k % PPP) of the first gather round of tile t, if that round has such a plane
    auto issue_round0_piece = [&](int t, int k) {
        const int pp = k / PPP, j = k % PPP;
        if (pp > cnt0) return;
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        int row;
        if (pp == 0) row = bstart + lrow0 + min(j < 4 ? 4 * j + q : lane >> 2, rows - 1);
        else row = j < 4 ? ix[pp - 1 < NBR ? pp - 1 : 0][j < 4 ? j : 0] : ixt[KG > 8 ? (pp - 1 < NBR ? pp - 1 : 0) : 0];
        issue_piece(pp, j, row);
    };
    // fragments of plane pp: NM float4 per lane (row i; read m = features 16 m + 4 q .. + 3)
    auto read_plane = [&](int pp, f32x4v (&f)[NM]) {
        const float* b = pl + pp * PLANE_F;
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            if (m < 4) f[m] = *reinterpret_cast<const f32x4v*>(b + i * 64 + (((4 * m + q) ^ i) << 2));
            else f[m] = *reinterpret_cast<const f32x4v*>(b + WT * 64 + i * 16 + ((q ^ swzt) << 2));
        }
    };

    // ---- prologue: the first tile's own rows, its indices and the bucket's weights (slot 0: W_self, slot 1: W_rel) are
    // requested together, the neighbours' rows as soon as the indices are here, and the weights go to LDS while those
    // travel: two memory round trips, not five
    int tc = t0, tn = t0 + tstep, tn2 = 0;               // current tile, next (rows in flight), the one after (indices)
    if (tc < t_end) {                                    // the first tile's own rows need no indices: they leave first
#pragma unroll
        for (int k = 0; k < PPP; ++k) issue_round0_piece(tc, k);
        if (deg > 0) load_idx(tc, 0);
    }

    const int w_rel = deg > 0 ? 2 * (deg - 1) : 20, w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    constexpr int NE = (WMAT_F / 4 + 511) / 512;
    float4 wa[NE], wb[NE];
    {
        // weights straight from the layer's [64][f_in] matrices into fragment order (no separate re-pack launch):
        //   image[mat][fb][m][lane][s] = W[16 fb + (lane & 15)][16 m + 4 (lane >> 4) + s], zero beyond f_in
        const float* W0 = P.w[w_self];
        const float* W1 = P.w[w_rel];
        const bool vec = (f_in & 3) == 0;
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int o = min(tid + 512 * e, WMAT_F / 4 - 1);
            const int ln = o & 63, fm = o >> 6;                  // fm = fb * NM + m
            const int row = 16 * (fm / NM) + (ln & 15), k = 16 * (fm % NM) + 4 * (ln >> 4);
            if (vec) {
                const bool ok = k < f_in;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
                wa[e] = ok ? *reinterpret_cast<const float4*>(W0 + row * f_in + k) : z;
                wb[e] = ok ? *reinterpret_cast<const float4*>(W1 + row * f_in + k) : z;
            } else {
                float va[4], vb[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int kk = min(k + u, f_in - 1);
                    va[u] = W0[row * f_in + kk];
                    vb[u] = W1[row * f_in + kk];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (k + u >= f_in) va[u] = vb[u] = 0.f;
                }
                wa[e] = make_float4(va[0], va[1], va[2], va[3]);
                wb[e] = make_float4(vb[0], vb[1], vb[2], vb[3]);
            }
        }
    }
    f32x4v bias[4];
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int col = 16 * fb + 4 * q + r;
            bias[fb][r] = P.b[w_self][col] + (deg > 0 ? P.b[w_rel][col] : 0.f);
        }
    }
    if (tc < t_end) {                                    // the indices are here: the neighbours' rows
#pragma unroll
        for (int k = PPP; k < NPL * PPP; ++k) issue_round0_piece(tc, k);
        if (deg > 0 && tn < t_end) load_idx(tn, 0);
    }
    {
        float4* d0 = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int e = 0; e < NE; ++e) {
            const int o = min(tid + 512 * e, WMAT_F / 4 - 1);        // (the clamped duplicates rewrite the same values)
            d0[o] = wa[e];
            d0[WMAT_F / 4 + o] = wb[e];
        }
    }
    __syncthreads();                                 // the weights are in LDS: the only barrier in front of the loop
    tn2 = tn + tstep;
    // The two waves of a SIMD (w and w + 4) run the same program; started together they stay in lockstep and meet the
    // matrix pipe, the LDS reads and the stores together.  Half a tile of delay for waves 4..7 puts one wave's
    // read / epilogue phases beside the other's MFMA phase (MI355X_MICROARCH.md, "try a stagger").
    if (wave >= 4 && JOVA_WAVE_STAGGER > 0) __builtin_amdgcn_s_sleep(JOVA_WAVE_STAGGER);

    float ssum[4][4], ssq[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) ssum[a][b] = ssq[a][b] = 0.f;
    const float* wl0 = lds + lane * 4;
    const float* wl1 = lds + WMAT_F + lane * 4;
    f32x4v prev[4];                                  // results of the previous tile, stored during this tile's MFMA phase
    float* prev_row = nullptr;
#pragma unroll
    for (int fb = 0; fb < 4; ++fb) prev[fb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#ifdef JOVA_WAVE_PROFILE
    unsigned long long pf_wait = 0, pf_read = 0, pf_mfma = 0, pf_epi = 0;
    const unsigned long long pf_start = __builtin_readcyclecounter();
#endif
    int n_done = 0;
    for (; tc < t_end; tc = tn, tn = tn2, tn2 += tstep, ++n_done) {
        const int t = tc;
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        WPROF(p0);
        // this tile's rows landed: they were requested during the previous tile's MFMA phase
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);
        asm volatile("" ::: "memory");
        WPROF(p1);
        f32x4v xf[NM], sf[NM];
        read_plane(0, xf);
#pragma unroll
        for (int m = 0; m < NM; ++m) sf[m] = (f32x4v){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int p = 0; p < NBR; ++p) {
            if (p < deg) {
                f32x4v f[NM];
                read_plane(1 + p, f);
#pragma unroll
                for (int m = 0; m < NM; ++m) sf[m] += f[m];
            }
        }
        for (int nb0 = NBR; nb0 < deg; nb0 += NBR) {               // further gather rounds of a high-degree tile
            load_idx(t, nb0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the planes' previous contents have been read
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (nb0 + p < deg) {
#pragma unroll
                    for (int j = 0; j < PPP; ++j) issue_piece(1 + p, j, j < 4 ? ix[p][j < 4 ? j : 0] : ixt[KG > 8 ? p : 0]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (nb0 + p < deg) {
                    f32x4v f[NM];
                    read_plane(1 + p, f);
#pragma unroll
                    for (int m = 0; m < NM; ++m) sf[m] += f[m];
                }
            }
            if (nb0 + NBR >= deg && tn < t_end) load_idx(tn, 0);            // (the registers held this tile's later rounds)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment of this tile is in registers
        WPROF(p2);
        // ---- MFMA phase; in its shadow: the next tile's rows into the (now free) planes, then the indices of the
        // tile after it, and the previous tile's four stores
        const bool nxt = tn < t_end;
        const bool nxt2 = tn2 < t_end && deg > 0 && deg <= NBR;
        auto side = [&](int k) {
            if (nxt && k < NPL * PPP) issue_round0_piece(tn, k);
            if (k == NS - 1 && nxt2) load_idx(tn2, 0);
            if (k >= NS - 4 && prev_row != nullptr)
                *reinterpret_cast<f32x4v*>(prev_row + 16 * (k - (NS - 4))) = prev[k - (NS - 4)];
        };
        f32x4v acc[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) acc[fb] = bias[fb];
        if (deg > 0) wave_mfma_tile<NM, true>(wl0, wl1, xf, sf, acc, side);
        else wave_mfma_tile<NM, false>(wl0, wl1, xf, sf, acc, side);
#ifdef JOVA_WAVE_PROFILE
        asm volatile("s_nop 0" ::"v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]));     // the MFMAs have retired
#endif
        WPROF(p4);
        // epilogue: lane (i, q) holds output features 16 fb + 4 q .. + 3 of atom i.  Rows past the bucket's end were
        // computed from the clamped (= last) row and store that row's values again.
        const bool live = i < rows;
        prev_row = out + (size_t)(bstart + lrow0 + min(i, rows - 1)) * 64 + 4 * q;
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
            f32x4v v = acc[fb];
            if (relu) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (STATS) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = live ? v[r] : 0.f;
                    ssum[fb][r] += u;
                    ssq[fb][r] += u * u;
                }
            }
            prev[fb] = v;
        }
#ifdef JOVA_WAVE_PROFILE
        const unsigned long long p5 = __builtin_readcyclecounter();
        pf_wait += p1 - p0; pf_read += p2 - p1; pf_mfma += p4 - p2; pf_epi += p5 - p4;
#endif
    }
    if (prev_row != nullptr) {
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) *reinterpret_cast<f32x4v*>(prev_row + 16 * fb) = prev[fb];
    }
#ifdef JOVA_WAVE_PROFILE
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_wave_prof + (blockIdx.x * 8 + wave) * 8;
        o[0] = pf_wait; o[1] = pf_read; o[2] = pf_start - pf_entry; o[3] = pf_mfma; o[4] = pf_epi;
        if (wave == 0) {                             // wall-clock (100 MHz) entry / exit of the workgroup's first wave
            g_wave_prof[256 * 64 + blockIdx.x * 2] = pf_wall0;
            g_wave_prof[256 * 64 + blockIdx.x * 2 + 1] = wall_clock64();
        }
        o[5] = __builtin_readcyclecounter() - pf_start; o[6] = n_done; o[7] = deg;
    }
#endif
    if (STATS) {
        // per-lane fp32 partials (a few dozen atoms each) -> fp64, over the 16 atom lanes, over the 8 waves, one atomic
        // per column and workgroup
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();                             // every wave is done with its planes
        double* red = reinterpret_cast<double*>(lds + 2 * WMAT_F);           // [8 waves][2][64]
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // (fp32 over the 16 atom lanes: each partial covers <= ~10 atoms, 160 values in all; fp64 from here on)
                float a = ssum[fb][r], b = ssq[fb][r];
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) {
                    a += __shfl_xor(a, o, 64);
                    b += __shfl_xor(b, o, 64);
                }
                if (i == 0) {
                    red[(wave * 2 + 0) * 64 + 16 * fb + 4 * q + r] = (double)a;
                    red[(wave * 2 + 1) * 64 + 16 * fb + 4 * q + r] = (double)b;
                }
            }
        }
        __syncthreads();
        if (tid < 128) {
            double a = 0.0;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[w * 128 + tid];
            atomicAdd(&stats[tid], a);               // stats = [2][64]: sums, then sums of squares
        }
    }
}

// ------------------------------------------------------------------------------------------------ backward, wave-private form
// The forward's organisation (a wave owns 16-atom tiles, rows by LDS-DMA into wave-owned planes, no workgroup barrier
// in the loop, the memory traffic of the neighbouring tiles issued in the MFMA shadow) applied to the backward with dZ
// handed in (the consumer applied the ReLU mask).  Per tile, on v_mfma_f32_16x16x4_f32:
//   phase A (only with dx):  [G | T]^T = [W_rel | W_self]^T dZ^T      128 MFMAs; the transposed weights sit in LDS in
//            fragment order (A operand), dZ is the B operand read atom-major as in the forward, so a lane ends with four
//            consecutive input features of one atom: 16-byte stores.  G -> gbuf (read by the neighbours in
//            graphconv_bwd_dx_v4_kernel), T -> dx.
//   phase B: dW_self += dZ^T X, dW_rel += dZ^T S                      2 x 16 x (f_in / 16) MFMAs; both operands are read
//            from the planes "transposed" (lane = (column, atom quad): one ds_read_b32 per fragment register), the
//            neighbour sum S is formed while reading.  The 2 x 64 x f_in accumulators (128 / 160 registers) stay in the
//            wave for all its tiles: a workgroup works on ONE degree bucket (WavePlan), so there is one reduction over
//            the eight waves (through the then idle planes) and one atomic per weight element and workgroup at the end.
// The planes are free once phase B's fragments are in registers: the next tile's DMA (own rows, neighbours, dZ) is
// issued between phase B's MFMA groups, the stores of G in phase A's second half, those of T in phase B.
template <int KG, bool DX, int NBR>
__global__ void __launch_bounds__(512, 1)
graphconv_bwd_wave_kernel(const float* __restrict__ x, int n_atoms, int ldx, int f_in, GcParams P,
                          const float* __restrict__ dz, GcGrads Gd, float* __restrict__ gbuf, float* __restrict__ dx,
                          float* __restrict__ dwp, WavePlan plan) {
    constexpr int NFB = KG / 2;                      // 16-feature blocks of the input rows
    constexpr int NPL = 1 + NBR;                     // x planes: own rows + NBR neighbour planes per gather round
    constexpr int PPP = KG > 8 ? 5 : 4;              // DMA pieces per x plane
    constexpr int NPC = NPL * PPP + 4;               // ... per tile (the dZ plane has four)
    constexpr int TAIL_F = KG > 8 ? WT * 16 : 0;
    constexpr int PLANE_F = WT * 64 + TAIL_F;
    constexpr int WAVE_F = NPL * PLANE_F + WT * 64;  // floats of a wave's planes (x planes, then the dZ plane)
    constexpr int WMAT_F = DX ? 4 * 4 * 256 : 0;     // the transposed W_rel image [fb][m][lane][s]
    constexpr int ACC_F = 16 * NFB * 64;             // floats of one wave's accumulator image of ONE weight matrix
    static_assert(PPP + 4 <= 12 && NBR * PPP <= 15, "every DMA piece of a tile needs a side slot in phase B");
    static_assert(4 * ACC_F + 8 * 64 <= 8 * WAVE_F, "the reduction image must fit in the planes");
    static_assert(!DX || KG == 8, "dx is produced for 64-float rows");
    __shared__ __attribute__((aligned(16))) float lds[WMAT_F + 8 * WAVE_F];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
#ifdef JOVA_WAVE_PROFILE
    const unsigned long long pw_entry = wall_clock64();
#endif

    // ---- this workgroup's bucket and its chunk of the bucket's tiles (as in the forward)
    int kb = 0;
#pragma unroll
    for (int k = 1; k < NB; ++k) kb += ((int)blockIdx.x >= plan.first[k]) ? 1 : 0;
    const int deg = NB - 1 - kb;
    const int wg0 = plan.first[kb], nwg = plan.first[kb + 1] - wg0;
    int bstart = 0, bsize = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        if (d == deg) { bstart = plan.start[d]; bsize = plan.size[d]; }
    }
    // XCD x (= blockIdx % 8) owns the x-th share of the bucket's tiles (shares in proportion to the XCD's workgroups of
    // this bucket); inside the share the tiles are dealt round-robin to ALL its waves, so at any moment the XCD's waves
    // of every bucket work at about the same relative position of their buckets: the rows one bucket streams are the
    // neighbour rows another gathers, and they meet in that XCD's L2 (a contiguous chunk per workgroup let the buckets
    // drift apart in time; measured: profiles/r02_pmc_graph.json).
    int before = 0, mine = 0, rank = 0;
    {
        const int xcd = blockIdx.x & 7, j = (int)blockIdx.x - wg0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j0 = (k - wg0) & 7;
            const int cnt = j0 < nwg ? (nwg - j0 + 7) >> 3 : 0;
            before += k < xcd ? cnt : 0;
            mine = k == xcd ? cnt : mine;
        }
        rank = (j - ((xcd - wg0) & 7)) >> 3;
    }
    const int tiles = (bsize + WT - 1) / WT;
    const int t_begin = (int)((long long)tiles * before / nwg) + rank * 8;
    const int t_end = (int)((long long)tiles * (before + mine) / nwg);
    const int tstep = mine * 8;
    const int t0 = t_begin + wave;

    float* const pl = lds + WMAT_F + wave * WAVE_F;
    float* const zpl = pl + NPL * PLANE_F;
    const unsigned pl_addr = lds_byte_addr(pl);
    const int* adj_d = deg > 0 ? P.adj[deg - 1] : nullptr;
    const unsigned nmax = (unsigned)(n_atoms - 1);
    const int cnt0 = min(deg, NBR);

    int ix[NBR][4];
    int ixt[KG > 8 ? NBR : 1];
    auto load_idx = [&](int t, int nb0) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
#pragma unroll
        for (int p = 0; p < NBR; ++p) {
            if (nb0 + p < deg) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int r = min(4 * j + q, rows - 1);
                    ix[p][j] = adj_d[(unsigned)((lrow0 + r) * deg + nb0 + p)];
                }
                if (KG > 8) {
                    const int r = min(lane >> 2, rows - 1);
                    ixt[p] = adj_d[(unsigned)((lrow0 + r) * deg + nb0 + p)];
                }
            }
        }
    };
    // (32-bit byte offsets from the uniform base pointers: the host checks n_atoms * ldx * 4 < 2^32)
    const unsigned ldx4 = (unsigned)ldx * 4u;
    const unsigned ch16 = (unsigned)(((lane & 15) ^ q) << 4);          // byte offset of chunk (lane & 15) ^ q
    const unsigned cht = 256u + (unsigned)((((lane & 3) ^ ((4 - (lane >> 4)) & 3))) << 4);
    auto issue_piece = [&](int pp, int j, int row) {            // x plane pp, piece j
        if (j < 4) {
            // row 4 j + q takes chunk (lane & 15) ^ (4 j + q) = ((lane & 15) ^ q) ^ 4 j
            dma16_m0_s(x, min((unsigned)row, nmax) * ldx4 + (ch16 ^ (unsigned)(j << 6)), pl_addr + (pp * PLANE_F + j * 256) * 4);
        } else {
            dma16_m0_s(x, min((unsigned)row, nmax) * ldx4 + cht, pl_addr + (pp * PLANE_F + WT * 64) * 4);
        }
    };
    // piece k of the first gather round of tile t: x planes first, then the four pieces of the dZ plane
    auto issue_round0_piece = [&](int t, int k) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        if (k >= NPL * PPP) {
            const int j = k - NPL * PPP;
            const unsigned row = (unsigned)(bstart + lrow0 + min(4 * j + q, rows - 1));
            dma16_m0_s(dz, min(row, nmax) * 256u + (ch16 ^ (unsigned)(j << 6)), pl_addr + (NPL * PLANE_F + j * 256) * 4);
            return;
        }
        const int pp = k / PPP, j = k % PPP;
        if (pp > cnt0) return;
        int row;
        if (pp == 0) row = bstart + lrow0 + min(j < 4 ? 4 * j + q : lane >> 2, rows - 1);
        else row = j < 4 ? ix[pp - 1 < NBR ? pp - 1 : 0][j < 4 ? j : 0] : ixt[KG > 8 ? (pp - 1 < NBR ? pp - 1 : 0) : 0];
        issue_piece(pp, j, row);
    };
    // "transposed" fragments of a plane: f[fb][ks] = plane[atom 4 ks + q][feature 16 fb + i]
    // (main image: chunk c = 4 fb + (i >> 2) of row r = 4 ks + q sits at position c ^ r = 4 (fb ^ ks) + ((i >> 2) ^ q):
    //  ONE lane-dependent offset, everything else is an instruction immediate)
    const int tsw = (i >> 2);
    const int lane_t = q * 64 + ((tsw ^ q) << 2) + (i & 3);
    const int lane_tt = q * 16 + (i & 3);
    auto read_plane_t = [&](const float* b, float (&f)[NFB][4], bool add) {
        const float* bl = b + lane_t;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                float v;
                if (fb < 4) v = bl[ks * 256 + ((fb ^ ks) << 4)];
                else v = b[WT * 64 + ks * 64 + lane_tt + ((tsw ^ ((4 - ks) & 3)) << 2)];
                f[fb][ks] = add ? f[fb][ks] + v : v;
            }
            // (adding a plane: one atom quad at a time, so that only NFB loaded values wait for their add)
            if (add) __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- prologue: weight loads, the first tile's indices and its index-free pieces (own rows, dZ) leave together; the
    // neighbours' pieces follow as soon as the indices are here (in-order counter: everything but the PPP + 4 pieces)
    int tc = t0, tn = t0 + tstep, tn2 = 0;
    const int w_rel = deg > 0 ? 2 * (deg - 1) : 20, w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    float4 wv[2];
    if (DX && deg > 0) {
        // image[fb][m][lane][s] = W_rel[16 m + 4 (lane >> 4) + s][16 fb + (lane & 15)]
        const float* W = P.w[w_rel];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int o = tid + 512 * e;
            const int ln = o & 63, fm = o >> 6;
            const int row = 16 * (fm & 3) + 4 * (ln >> 4), col = 16 * (fm >> 2) + (ln & 15);
            wv[e] = make_float4(W[(row + 0) * f_in + col], W[(row + 1) * f_in + col], W[(row + 2) * f_in + col],
                                W[(row + 3) * f_in + col]);
        }
    }
    if (tc < t_end) {
        if (deg > 0) load_idx(tc, 0);
#pragma unroll
        for (int k = 0; k < PPP; ++k) issue_round0_piece(tc, k);
#pragma unroll
        for (int k = NPL * PPP; k < NPC; ++k) issue_round0_piece(tc, k);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70 | (PPP + 4));
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = PPP; k < NPL * PPP; ++k) issue_round0_piece(tc, k);
        if (deg > 0 && tn < t_end) load_idx(tn, 0);
    }
    if (DX && deg > 0) {
        float4* d0 = reinterpret_cast<float4*>(lds);
#pragma unroll
        for (int e = 0; e < 2; ++e) d0[tid + 512 * e] = wv[e];
    }
    __syncthreads();
    tn2 = tn + tstep;
    if (wave >= 4 && JOVA_WAVE_STAGGER > 0) __builtin_amdgcn_s_sleep(JOVA_WAVE_STAGGER);

    f32x4v accS[4][NFB], accR[4][NFB];               // dW_self / dW_rel [16 nb + 4 q + r][16 fb + i]
    float dbacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        dbacc[nb] = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) accS[nb][fb] = accR[nb][fb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    const float* wl0 = lds + lane * 4;               // W_rel^T image
#ifdef JOVA_WAVE_PROFILE
    unsigned long long pf[6] = {0, 0, 0, 0, 0, 0};
    const unsigned long long pw_loop = wall_clock64();
    const unsigned long long pf_start = __builtin_readcyclecounter();
    int n_done = 0;
#endif

    for (; tc < t_end; tc = tn, tn = tn2, tn2 += tstep) {
        const int t = tc;
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        WPROF(p0);
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_waitcnt(0x0F70);          // this tile's planes landed
        asm volatile("" ::: "memory");
        WPROF(p1);
        f32x4v accG[4];
        float* grow = nullptr;
        if (DX && deg > 0) {
            // ---- phase A: G^T = W_rel^T dZ^T (the neighbours' share of dx; the own share T = dZ W_self is formed by
            // graphconv_bwd_dx_wave_kernel, which is memory-bound and has the matrix pipe to spare)
            f32x4v zf[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) zf[m] = *reinterpret_cast<const f32x4v*>(zpl + i * 64 + (((4 * m + q) ^ i) << 2));
            grow = gbuf + (size_t)(bstart + lrow0 + min(i, rows - 1)) * 64 + 4 * q;
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) accG[fb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            f32x4v w[2][4];
#pragma unroll
            for (int fb = 0; fb < 4; ++fb) w[0][fb] = *reinterpret_cast<const f32x4v*>(wl0 + (fb * 4) * 256);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int c = m & 1, nx = c ^ 1;
                if (m + 1 < 4) {
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb)
                        w[nx][fb] = *reinterpret_cast<const f32x4v*>(wl0 + (fb * 4 + m + 1) * 256);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
#pragma unroll
                    for (int fb = 0; fb < 4; ++fb)
                        accG[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][fb][s], zf[m][s], accG[fb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
#ifdef JOVA_WAVE_PROFILE
        if (DX && deg > 0) asm volatile("s_nop 0" ::"v"(accG[0]), "v"(accG[1]), "v"(accG[2]), "v"(accG[3]));
#endif
        WPROF(p2);
        // ---- phase B1: dW_self += dZ^T X.  Once dZ^T and X^T are in registers the own plane and the dZ plane are free:
        // the next tile's pieces for them are issued between the MFMA groups, then T's four stores.
        float zt[4][4], xs[NFB][4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const bool live = 4 * ks + q < rows;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const float v = zpl[lane_t + ks * 256 + ((nb ^ ks) << 4)];
                zt[nb][ks] = live ? v : 0.f;             // rows past the bucket's end repeat its last row: no contribution
            }
        }
        read_plane_t(pl, xs, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) dbacc[nb] += (zt[nb][0] + zt[nb][1]) + (zt[nb][2] + zt[nb][3]);
        WPROF(p3);
        const bool nxt = tn < t_end;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int k = 4 * ks + nb;
                if (k < PPP) { if (nxt) issue_round0_piece(tn, k); }                         // own plane
                else if (k < PPP + 4) { if (nxt) issue_round0_piece(tn, NPL * PPP + k - PPP); }   // dZ plane
                else if (DX && k >= 12 && grow != nullptr) *reinterpret_cast<f32x4v*>(grow + 16 * (k - 12)) = accG[k - 12];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    accS[nb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nb][ks], xs[fb][ks], accS[nb][fb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef JOVA_WAVE_PROFILE
        asm volatile("s_nop 0" ::"v"(accS[3][NFB - 1]));
        const unsigned long long p4 = __builtin_readcyclecounter();
        pf[0] += p1 - p0; pf[1] += p2 - p1; pf[2] += p3 - p2; pf[3] += p4 - p3;
        ++n_done;
#endif
        if (deg == 0) continue;                      // (workgroup-uniform)

        // ---- phase B2: dW_rel += dZ^T S, S = sum of the neighbour rows (formed while reading, plane by plane)
        float xn[NFB][4];
        read_plane_t(pl + PLANE_F, xn, false);
        if (NBR > 1 && deg > 1) {
            __builtin_amdgcn_sched_barrier(0);
            read_plane_t(pl + (NBR > 1 ? 2 : 1) * PLANE_F, xn, true);
        }
        for (int nb0 = NBR; nb0 < deg; nb0 += NBR) {               // further gather rounds of a high-degree tile
            load_idx(t, nb0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the planes' previous contents have been read
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (nb0 + p < deg) {
#pragma unroll
                    for (int j = 0; j < PPP; ++j) issue_piece(1 + p, j, j < 4 ? ix[p][j < 4 ? j : 0] : ixt[KG > 8 ? p : 0]);
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (nb0 + p < deg) {
                    __builtin_amdgcn_sched_barrier(0);
                    read_plane_t(pl + (1 + p) * PLANE_F, xn, true);
                }
            }
            if (nb0 + NBR >= deg && tn < t_end) load_idx(tn, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment of this tile is in registers
        WPROF(p5);
        const bool nxt2 = tn2 < t_end && deg <= NBR;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                const int k = 4 * ks + nb;
                if (k < NBR * PPP) { if (nxt) issue_round0_piece(tn, PPP + k); }            // neighbour planes
                if (k == 15 && nxt2) load_idx(tn2, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    accR[nb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nb][ks], xn[fb][ks], accR[nb][fb], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef JOVA_WAVE_PROFILE
        asm volatile("s_nop 0" ::"v"(accR[3][NFB - 1]));
        const unsigned long long p6 = __builtin_readcyclecounter();
        pf[4] += p5 - p4; pf[5] += p6 - p5;
#endif
    }
#ifdef JOVA_WAVE_PROFILE
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_wave_prof + (blockIdx.x * 8 + wave) * 8;
        o[0] = pf[0]; o[1] = pf[1]; o[2] = pf[2]; o[3] = pf[3]; o[4] = pf[4];
        o[5] = pf[5]; o[6] = n_done; o[7] = deg;
    }
    const unsigned long long pw_end = wall_clock64();
#endif

    // ---- reduction over the eight waves (through the planes), then one atomic per weight element and workgroup
    __builtin_amdgcn_s_waitcnt(0x0F70);
    float* const red = lds + WMAT_F;                 // [4 waves][ACC_F], then [8 waves][64] column sums of dZ
    float* const dbred = red + 4 * ACC_F;
    float* const ws = dwp ? dwp + (long long)w_self * 64 * DWP_LD : Gd.dw[w_self];
    float* const wr = dwp ? dwp + (long long)w_rel * 64 * DWP_LD : Gd.dw[w_rel];
    const int ldw = dwp ? DWP_LD : f_in;
#pragma unroll
    for (int mat = 0; mat < 2; ++mat) {
        if (mat == 1 && deg == 0) break;             // (workgroup-uniform)
        f32x4v (&acc)[4][NFB] = mat == 0 ? accS : accR;
        __syncthreads();                             // the planes (or the previous matrix's image) are idle
        if (wave >= 4) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    *reinterpret_cast<f32x4v*>(red + (wave - 4) * ACC_F + ((nb * NFB + fb) * 64 + lane) * 4) = acc[nb][fb];
        }
        __syncthreads();
        if (wave < 4) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb) {
                    f32x4v* p = reinterpret_cast<f32x4v*>(red + wave * ACC_F + ((nb * NFB + fb) * 64 + lane) * 4);
                    *p = *p + acc[nb][fb];
                }
        }
        if (mat == 0) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                float s = dbacc[nb];
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (q == 0) dbred[wave * 64 + 16 * nb + i] = s;
            }
        }
        __syncthreads();
        float* const dst = mat == 0 ? ws : wr;
        for (int e = tid; e < ACC_F / 4; e += 512) {
            // float4 e of the image: (nb, fb) = e / 64, lane' = e % 64 -> rows o = 16 nb + 4 q' + r, column 16 fb + i'
            const float4 a = *reinterpret_cast<const float4*>(red + 4 * e);
            const float4 b = *reinterpret_cast<const float4*>(red + ACC_F + 4 * e);
            const float4 c = *reinterpret_cast<const float4*>(red + 2 * ACC_F + 4 * e);
            const float4 d = *reinterpret_cast<const float4*>(red + 3 * ACC_F + 4 * e);
            const int ln = e & 63, bf = e >> 6;
            const int o = 16 * (bf / NFB) + 4 * (ln >> 4), k = 16 * (bf % NFB) + (ln & 15);
            if (k < f_in) {
                atomicAdd(&dst[(o + 0) * ldw + k], (a.x + b.x) + (c.x + d.x));
                atomicAdd(&dst[(o + 1) * ldw + k], (a.y + b.y) + (c.y + d.y));
                atomicAdd(&dst[(o + 2) * ldw + k], (a.z + b.z) + (c.z + d.z));
                atomicAdd(&dst[(o + 3) * ldw + k], (a.w + b.w) + (c.w + d.w));
            }
        }
        if (mat == 0 && tid < 64) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += dbred[w * 64 + tid];
            atomicAdd(&Gd.db[w_self][tid], s);
            if (deg > 0) atomicAdd(&Gd.db[w_rel][tid], s);
        }
    }
#ifdef JOVA_WAVE_PROFILE
    __builtin_amdgcn_s_waitcnt(0x0F70);
    if (lane == 0 && blockIdx.x < 256 && wave == 0) {
        unsigned long long* o = g_wave_prof + 256 * 64 + blockIdx.x * 2;
        o[0] = (pw_entry & 0xFFFFFFFFull) | ((pw_loop & 0xFFFFFFFFull) << 32);
        o[1] = (pw_end & 0xFFFFFFFFull) | ((wall_clock64() & 0xFFFFFFFFull) << 32);
    }
#endif
}

// ------------------------------------------------------------------------------------------------ backward, dW only, waves by matrix
// 80-float rows (75 -> 64): the 2 x 64 x 80 weight-gradient accumulators are 160 registers per lane, which with the
// fragments does not fit the 256 a wave may hold when two share a SIMD.  Here the two waves of a SIMD (w and w + 4)
// work on the SAME tiles and each carries ONE matrix: waves 0..3 dW_self += dZ^T X (planes: own rows, dZ), waves 4..7
// dW_rel += dZ^T S (planes: four neighbours, dZ) -- 80 accumulator registers each, the same 2 x 16 x 5 MFMAs per tile and
// SIMD, no workgroup barrier before the final reduction, and with four neighbour planes only atoms of degree > 4 need
// a second gather round.  A tile's neighbour indices are ONE load of its contiguous adjacency block (16 x degree ints,
// up to three registers) and cross-lane reads, instead of one load per (plane, row group).
template <int KG>
__global__ void __launch_bounds__(512, 1)
graphconv_bwd_split_kernel(const float* __restrict__ x, int n_atoms, int ldx, int f_in, GcParams P,
                           const float* __restrict__ dz, GcGrads Gd, float* __restrict__ dwp, WavePlan plan) {
    constexpr int NFB = KG / 2;
    constexpr int NBR = 4;
    constexpr int PPP = KG > 8 ? 5 : 4;
    constexpr int TAIL_F = KG > 8 ? WT * 16 : 0;
    constexpr int PLANE_F = WT * 64 + TAIL_F;
    constexpr int SELF_F = PLANE_F + WT * 64;            // own rows, dZ
    constexpr int REL_F = NBR * PLANE_F + WT * 64;       // neighbour planes, dZ
    constexpr int ACC_F = 16 * NFB * 64;
    static_assert(4 * ACC_F + 4 * 64 <= 4 * (SELF_F + REL_F), "the reduction image must fit in the planes");
    __shared__ __attribute__((aligned(16))) float lds[4 * (SELF_F + REL_F)];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    const bool rel = wave >= 4;
    const int pair = wave & 3;

    int kb = 0;
#pragma unroll
    for (int k = 1; k < NB; ++k) kb += ((int)blockIdx.x >= plan.first[k]) ? 1 : 0;
    const int deg = NB - 1 - kb;
    const int wg0 = plan.first[kb], nwg = plan.first[kb + 1] - wg0;
    int bstart = 0, bsize = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        if (d == deg) { bstart = plan.start[d]; bsize = plan.size[d]; }
    }
    int before = 0, mine = 0, rank = 0;
    {
        const int xcd = blockIdx.x & 7, j = (int)blockIdx.x - wg0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int j0 = (k - wg0) & 7;
            const int cnt = j0 < nwg ? (nwg - j0 + 7) >> 3 : 0;
            before += k < xcd ? cnt : 0;
            mine = k == xcd ? cnt : mine;
        }
        rank = (j - ((xcd - wg0) & 7)) >> 3;
    }
    const int tiles = (bsize + WT - 1) / WT;
    const int t_begin = (int)((long long)tiles * before / nwg) + rank * 4;
    const int t_end = (rel && deg == 0) ? 0 : (int)((long long)tiles * (before + mine) / nwg);
    const int tstep = mine * 4;

    float* const pl = lds + (rel ? 4 * SELF_F + pair * REL_F : pair * SELF_F);
    float* const zpl = pl + (rel ? NBR * PLANE_F : PLANE_F);
    const unsigned pl_addr = lds_byte_addr(pl);
    const unsigned z_addr = lds_byte_addr(zpl);
    const int* adj_d = deg > 0 ? P.adj[deg - 1] : nullptr;
    const unsigned nmax = (unsigned)(n_atoms - 1);
    const int cnt0 = min(deg, NBR);
    const int nraw = (16 * deg + 63) >> 6;               // registers of a tile's adjacency block (<= 3)
    const unsigned ldx4 = (unsigned)ldx * 4u;
    const unsigned ch16 = (unsigned)(((lane & 15) ^ q) << 4);
    const unsigned cht = 256u + (unsigned)((((lane & 3) ^ ((4 - (lane >> 4)) & 3))) << 4);

    auto issue_x_piece = [&](int pp, int j, unsigned row) {
        if (j < 4) dma16_m0_s(x, min(row, nmax) * ldx4 + (ch16 ^ (unsigned)(j << 6)), pl_addr + (pp * PLANE_F + j * 256) * 4);
        else dma16_m0_s(x, min(row, nmax) * ldx4 + cht, pl_addr + (pp * PLANE_F + WT * 64) * 4);
    };
    auto issue_z_piece = [&](int t, int j) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        const unsigned row = (unsigned)(bstart + lrow0 + min(4 * j + q, rows - 1));
        dma16_m0_s(dz, min(row, nmax) * 256u + (ch16 ^ (unsigned)(j << 6)), z_addr + j * 1024);
    };
    auto issue_own_piece = [&](int t, int j) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        issue_x_piece(0, j, (unsigned)(bstart + lrow0 + min(j < 4 ? 4 * j + q : lane >> 2, rows - 1)));
    };
    // the tile's adjacency block: element e = (local row) * deg + p, register e >> 6, lane e & 63
    auto load_raw = [&](int t, int (&raw)[3]) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        const unsigned last = (unsigned)(rows * deg - 1);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            if (u < nraw) raw[u] = adj_d[(unsigned)(lrow0 * deg) + min((unsigned)(64 * u + lane), last)];
        }
    };
    auto pick = [&](const int (&raw)[3], int t, int r, int p) -> unsigned {        // neighbour p of local row r
        const int rows = min(WT, bsize - t * WT);
        const int e = min(r, rows - 1) * deg + p;
        int v = __shfl(raw[0], e & 63, 64);
        if (nraw > 1) { const int v1 = __shfl(raw[1], e & 63, 64); v = (e >> 6) == 1 ? v1 : v; }
        if (nraw > 2) { const int v2 = __shfl(raw[2], e & 63, 64); v = (e >> 6) == 2 ? v2 : v; }
        return (unsigned)v;
    };
    const int tsw = (i >> 2);
    const int lane_t = q * 64 + ((tsw ^ q) << 2) + (i & 3);
    const int lane_tt = q * 16 + (i & 3);
    auto read_plane_t = [&](const float* b, float (&f)[NFB][4], bool add) {
        const float* bl = b + lane_t;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int fb = 0; fb < NFB; ++fb) {
                float v;
                if (fb < 4) v = bl[ks * 256 + ((fb ^ ks) << 4)];
                else v = b[WT * 64 + ks * 64 + lane_tt + ((tsw ^ ((4 - ks) & 3)) << 2)];
                f[fb][ks] = add ? f[fb][ks] + v : v;
            }
        }
    };

    f32x4v acc[4][NFB];
    float dbacc[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        dbacc[nb] = 0.f;
#pragma unroll
        for (int fb = 0; fb < NFB; ++fb) acc[nb][fb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
    }
    int tc = t_begin + pair, tn = tc + tstep;
#ifdef JOVA_WAVE_PROFILE
    unsigned long long pf[3] = {0, 0, 0}, pf_mid = 0;
    int n_done = 0;
    const unsigned long long pw_entry = wall_clock64();
#endif
    if (!rel) {
        // ================================================================ waves 0..3: dW_self (and db)
        if (tc < t_end) {
#pragma unroll
            for (int j = 0; j < PPP; ++j) issue_own_piece(tc, j);
#pragma unroll
            for (int j = 0; j < 4; ++j) issue_z_piece(tc, j);
        }
        for (; tc < t_end; tc = tn, tn += tstep) {
            const int rows = min(WT, bsize - tc * WT);
            WPROF(p0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            WPROF(p1);
            float zt[4][4], xs[NFB][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bool live = 4 * ks + q < rows;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const float v = zpl[lane_t + ks * 256 + ((nb ^ ks) << 4)];
                    zt[nb][ks] = live ? v : 0.f;
                }
            }
            read_plane_t(pl, xs, false);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) dbacc[nb] += (zt[nb][0] + zt[nb][1]) + (zt[nb][2] + zt[nb][3]);
            WPROF(p2);
            const bool nxt = tn < t_end;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int k = 4 * ks + nb;
                    if (nxt && k < PPP) issue_own_piece(tn, k);
                    else if (nxt && k < PPP + 4) issue_z_piece(tn, k - PPP);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb)
                        acc[nb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nb][ks], xs[fb][ks], acc[nb][fb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef JOVA_WAVE_PROFILE
            asm volatile("s_nop 0" ::"v"(acc[3][NFB - 1]));
            const unsigned long long p3 = __builtin_readcyclecounter();
            pf[0] += p1 - p0; pf[1] += p2 - p1; pf[2] += p3 - p2; ++n_done;
#endif
        }
    } else {
        // ================================================================ waves 4..7: dW_rel
        int raw_c[3] = {0, 0, 0}, raw_n[3] = {0, 0, 0};
        unsigned ixn[NBR][4], ixtn[NBR];
        // rows (in the DMA's lane -> row map) of gather round nb0 of tile t, from its adjacency block
        const int eb[4] = {q * deg, (4 + q) * deg, (8 + q) * deg, (12 + q) * deg};
        const int ebt = (lane >> 2) * deg;
        auto rows_of = [&](const int (&raw)[3], int t, int nb0) {
            if (nraw == 1) {
                // degree <= 4 (the block is one register): all cross-lane reads are issued back to back, no select, no
                // wait in between (planes beyond the degree read a clamped element and are never issued)
                // (element numbers from per-lane constants: a 32-bit multiply is a quarter-rate instruction; rows past a
                // partial tile's end read its last element, they only need a valid row: their dZ is zeroed)
                const int last = min(WT, bsize - t * WT) * deg - 1;
#pragma unroll
                for (int p = 0; p < NBR; ++p) {
                    if (p < deg) {                      // (wave-uniform: a cross-lane read costs ~70 cycles here)
#pragma unroll
                        for (int j = 0; j < 4; ++j) ixn[p][j] = (unsigned)__shfl(raw[0], min(eb[j] + p, last), 64);
                        if (KG > 8) ixtn[p] = (unsigned)__shfl(raw[0], min(ebt + p, last), 64);
                    }
                }
                return;
            }
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (nb0 + p < deg) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) ixn[p][j] = pick(raw, t, 4 * j + q, nb0 + p);
                    if (KG > 8) ixtn[p] = pick(raw, t, lane >> 2, nb0 + p);
                }
            }
        };
        if (tc < t_end) {
            load_raw(tc, raw_c);
#pragma unroll
            for (int j = 0; j < 4; ++j) issue_z_piece(tc, j);
            rows_of(raw_c, tc, 0);
#pragma unroll
            for (int p = 0; p < NBR; ++p) {
                if (p < cnt0) {
#pragma unroll
                    for (int j = 0; j < PPP; ++j) issue_x_piece(p, j, j < 4 ? ixn[p][j < 4 ? j : 0] : ixtn[p]);
                }
            }
            if (tn < t_end) load_raw(tn, raw_n);
        }
        for (; tc < t_end; tc = tn, tn += tstep) {
            const int rows = min(WT, bsize - tc * WT);
            WPROF(p0);
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0x0F70);
            asm volatile("" ::: "memory");
            WPROF(p1);
            float zt[4][4], xn[NFB][4];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const bool live = 4 * ks + q < rows;
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const float v = zpl[lane_t + ks * 256 + ((nb ^ ks) << 4)];
                    zt[nb][ks] = live ? v : 0.f;
                }
            }
            read_plane_t(pl, xn, false);
            // (this wave has registers to spare: the planes are read together, one LDS round trip for all of them)
#pragma unroll
            for (int p = 1; p < NBR; ++p) {
                if (p < deg) read_plane_t(pl + p * PLANE_F, xn, true);
            }
            for (int nb0 = NBR; nb0 < deg; nb0 += NBR) {            // degree > 4: further gather rounds
                rows_of(raw_c, tc, nb0);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the planes' previous contents have been read
#pragma unroll
                for (int p = 0; p < NBR; ++p) {
                    if (nb0 + p < deg) {
#pragma unroll
                        for (int j = 0; j < PPP; ++j) issue_x_piece(p, j, j < 4 ? ixn[p][j < 4 ? j : 0] : ixtn[p]);
                    }
                }
                __builtin_amdgcn_s_waitcnt(0x0F70);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int p = 0; p < NBR; ++p) {
                    if (nb0 + p < deg) {
                        __builtin_amdgcn_sched_barrier(0);
                        read_plane_t(pl + p * PLANE_F, xn, true);
                    }
                }
            }
            const bool nxt = tn < t_end;
#ifdef JOVA_WAVE_PROFILE
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long p15 = __builtin_readcyclecounter();
            pf_mid += p15 - p1;
#endif
            if (nxt) rows_of(raw_n, tn, 0);                          // (the block of tile tn arrived a tile ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            WPROF(p2);
            // 16 MFMA groups; in front of group k: piece k and piece k + 16 of the next tile (4 of dZ, then 5 per plane)
            auto side = [&](int k) {
                if (!nxt) return;
                if (k < 4) { issue_z_piece(tn, k); return; }
                const int pp = (k - 4) / PPP, j = (k - 4) % PPP;
                if (pp < cnt0) issue_x_piece(pp, j, j < 4 ? ixn[pp < NBR ? pp : 0][j < 4 ? j : 0] : ixtn[pp < NBR ? pp : 0]);
            };
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    const int k = 4 * ks + nb;
                    side(k);
                    if (k + 16 < 4 + NBR * PPP) side(k + 16);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int fb = 0; fb < NFB; ++fb)
                        acc[nb][fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(zt[nb][ks], xn[fb][ks], acc[nb][fb], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef JOVA_WAVE_PROFILE
            asm volatile("s_nop 0" ::"v"(acc[3][NFB - 1]));
            const unsigned long long p3 = __builtin_readcyclecounter();
            pf[0] += p1 - p0; pf[1] += p2 - p1; pf[2] += p3 - p2; ++n_done;
#endif
#pragma unroll
            for (int u = 0; u < 3; ++u) raw_c[u] = raw_n[u];
            if (tn + tstep < t_end) load_raw(tn + tstep, raw_n);
        }
    }

#ifdef JOVA_WAVE_PROFILE
    const unsigned long long pw_end = wall_clock64();
    if (lane == 0 && blockIdx.x < 256) {
        unsigned long long* o = g_wave_prof + (blockIdx.x * 8 + wave) * 8;
        o[0] = pf[0]; o[1] = pf[1]; o[2] = rel; o[3] = pf[2]; o[4] = pw_end - pw_entry; o[5] = pf_mid; o[6] = n_done; o[7] = deg;
    }
#endif
    // ---- reduction over the four waves of each matrix (through the planes), one atomic per element and workgroup
    __builtin_amdgcn_s_waitcnt(0x0F70);
    const int w_rel = deg > 0 ? 2 * (deg - 1) : 20, w_self = deg == 0 ? 20 : 2 * (deg - 1) + 1;
    float* const red = lds;                               // [4 waves][ACC_F], then [4 waves][64] column sums of dZ
    float* const dbred = red + 4 * ACC_F;
    const int ldw = dwp ? DWP_LD : f_in;
#pragma unroll
    for (int mat = 0; mat < 2; ++mat) {
        if (mat == 1 && deg == 0) break;                  // (workgroup-uniform)
        __syncthreads();
        if (rel == (mat == 1)) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb)
#pragma unroll
                for (int fb = 0; fb < NFB; ++fb)
                    *reinterpret_cast<f32x4v*>(red + pair * ACC_F + ((nb * NFB + fb) * 64 + lane) * 4) = acc[nb][fb];
            if (mat == 0) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) {
                    float s = dbacc[nb];
                    s += __shfl_xor(s, 16, 64);
                    s += __shfl_xor(s, 32, 64);
                    if (q == 0) dbred[pair * 64 + 16 * nb + i] = s;
                }
            }
        }
        __syncthreads();
        float* const dst = dwp ? dwp + (long long)(mat == 0 ? w_self : w_rel) * 64 * DWP_LD : Gd.dw[mat == 0 ? w_self : w_rel];
        for (int e = tid; e < ACC_F / 4; e += 512) {
            const float4 a = *reinterpret_cast<const float4*>(red + 4 * e);
            const float4 b = *reinterpret_cast<const float4*>(red + ACC_F + 4 * e);
            const float4 c = *reinterpret_cast<const float4*>(red + 2 * ACC_F + 4 * e);
            const float4 d = *reinterpret_cast<const float4*>(red + 3 * ACC_F + 4 * e);
            const int ln = e & 63, bf = e >> 6;
            const int o = 16 * (bf / NFB) + 4 * (ln >> 4), k = 16 * (bf % NFB) + (ln & 15);
            if (k < f_in) {
                atomicAdd(&dst[(o + 0) * ldw + k], (a.x + b.x) + (c.x + d.x));
                atomicAdd(&dst[(o + 1) * ldw + k], (a.y + b.y) + (c.y + d.y));
                atomicAdd(&dst[(o + 2) * ldw + k], (a.z + b.z) + (c.z + d.z));
                atomicAdd(&dst[(o + 3) * ldw + k], (a.w + b.w) + (c.w + d.w));
            }
        }
        if (mat == 0 && tid < 64) {
            const float s = (dbred[tid] + dbred[64 + tid]) + (dbred[128 + tid] + dbred[192 + tid]);
            atomicAdd(&Gd.db[w_self][tid], s);
            if (deg > 0) atomicAdd(&Gd.db[w_rel][tid], s);
        }
    }
}

// ------------------------------------------------------------------------------------------------ dx of the wave-private backward
// dx[a] = dZ[a] W_self[deg(a)]  +  sum over a's neighbours n of G[n]      (G = dZ W_rel from graphconv_bwd_wave_kernel)
// Memory-bound (reads dZ and the gathered G rows, writes dx), so the 64 MFMAs per 16-atom tile of the own share ride
// along for free and T never exists in memory.  A wave owns a tile: dZ fragments and the neighbours' G rows come
// straight into registers (a lane holds four consecutive features of one atom, as in the forward's epilogue), many
// small workgroups (one degree bucket each, its transposed W_self in LDS) hide the gather latency by occupancy.
// Asymmetric adjacency (*sym_flag == 0): only the own share is written here, graphconv_bwd_dx_scatter_kernel adds the
// rest with float atomics.
__global__ void __launch_bounds__(256, 4)
graphconv_bwd_dx_wave_kernel(const float* __restrict__ dz, int n_atoms, GcParams P, const float* __restrict__ gbuf,
                             float* __restrict__ dx, const int* __restrict__ sym_flag, WavePlan plan, int tpw) {
    __shared__ __attribute__((aligned(16))) float wimg[4 * 4 * 256];
    __shared__ __attribute__((aligned(16))) float tile[4][WT * 64];       // per wave: the dZ tile, then the T tile
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 15, q = lane >> 4;
    int kb = 0;
#pragma unroll
    for (int k = 1; k < NB; ++k) kb += ((int)blockIdx.x >= plan.first[k]) ? 1 : 0;
    const int deg = NB - 1 - kb;
    int bstart = 0, bsize = 0;
#pragma unroll
    for (int d = 0; d < NB; ++d) {
        if (d == deg) { bstart = plan.start[d]; bsize = plan.size[d]; }
    }
    const int tiles = (bsize + WT - 1) / WT;
    const int t_first = ((int)blockIdx.x - plan.first[kb]) * 4 * tpw;
    const int t_last = min(tiles, t_first + 4 * tpw);
    const int* adj_d = deg > 0 ? P.adj[deg - 1] : nullptr;
    const unsigned nmax = (unsigned)(n_atoms - 1);
    float* const tl = tile[wave];
    const unsigned tl_addr = lds_byte_addr(tl);
    const unsigned ch16 = (unsigned)(((lane & 15) ^ q) << 4);
    // the wave's first dZ tile travels while the weights are staged
    auto issue_z = [&](int t) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned row = (unsigned)(bstart + lrow0 + min(4 * j + q, rows - 1));
            dma16_m0_s(dz, min(row, nmax) * 256u + (ch16 ^ (unsigned)(j << 6)), tl_addr + j * 1024);
        }
    };
    const int np = *sym_flag != 0 ? deg : 0;
    // indices of a tile in the gather layout: lane -> atom 4 j + q (j = 0..3), 16-byte chunk i of its row
    // (list positions 0..3 are all requested a tile ahead: a degree-3 or -4 tile, the common case of a bond graph,
    // starts its second pair of gathers without another index round trip)
    unsigned nb0[4], nb1[4], nb2[4], nb3[4];
    auto load_nb = [&](int t) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int lr = lrow0 + min(4 * j + q, rows - 1);
            nb0[j] = nb1[j] = nb2[j] = nb3[j] = 0;
            if (np > 0) nb0[j] = (unsigned)adj_d[(unsigned)(lr * deg)];
            if (np > 1) nb1[j] = (unsigned)adj_d[(unsigned)(lr * deg + 1)];
            if (np > 2) nb2[j] = (unsigned)adj_d[(unsigned)(lr * deg + 2)];
            if (np > 3) nb3[j] = (unsigned)adj_d[(unsigned)(lr * deg + 3)];
        }
    };
    // the wave's first dZ tile and its first indices travel while the weights are staged
    if (t_first + wave < t_last) {
        issue_z(t_first + wave);
        load_nb(t_first + wave);
    }
    {
        // image[fb][m][lane][s] = W_self[16 m + 4 (lane >> 4) + s][16 fb + (lane & 15)]
        const float* W = P.w[deg == 0 ? 20 : 2 * (deg - 1) + 1];
        float4 v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int o = tid + 256 * e;
            const int ln = o & 63, fm = o >> 6;
            const int row = 16 * (fm & 3) + 4 * (ln >> 4), col = 16 * (fm >> 2) + (ln & 15);
            v[e] = make_float4(W[(row + 0) * 64 + col], W[(row + 1) * 64 + col], W[(row + 2) * 64 + col], W[(row + 3) * 64 + col]);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) reinterpret_cast<float4*>(wimg)[tid + 256 * e] = v[e];
    }
    __syncthreads();
    const float* wl = wimg + lane * 4;
    bool first = true;
    for (int t = t_first + wave; t < t_last; t += 4) {
        const int lrow0 = t * WT, rows = min(WT, bsize - lrow0);
        int lr[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) lr[j] = lrow0 + min(4 * j + q, rows - 1);
        // this tile's dZ and indices were requested in front of the previous tile's four stores: those may still travel
        asm volatile("" ::: "memory");
        if (first) __builtin_amdgcn_s_waitcnt(0x0F70);
        else __builtin_amdgcn_s_waitcnt(0x0F74);
        asm volatile("" ::: "memory");
        first = false;
        // the neighbours' G rows of the first two list positions travel during the MFMAs
        f32x4v g0[4], g1[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            g0[j] = g1[j] = (f32x4v){0.f, 0.f, 0.f, 0.f};
            if (np > 0) g0[j] = *reinterpret_cast<const f32x4v*>(gbuf + (size_t)min(nb0[j], nmax) * 64 + 4 * i);
            if (np > 1) g1[j] = *reinterpret_cast<const f32x4v*>(gbuf + (size_t)min(nb1[j], nmax) * 64 + 4 * i);
        }
        f32x4v zf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) zf[m] = *reinterpret_cast<const f32x4v*>(tl + i * 64 + (((4 * m + q) ^ i) << 2));
        f32x4v acc[4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) acc[fb] = (f32x4v){0.f, 0.f, 0.f, 0.f};
        f32x4v w[2][4];
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) w[0][fb] = *reinterpret_cast<const f32x4v*>(wl + (fb * 4) * 256);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int c = m & 1, nx = c ^ 1;
            if (m + 1 < 4) {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb) w[nx][fb] = *reinterpret_cast<const f32x4v*>(wl + (fb * 4 + m + 1) * 256);
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int fb = 0; fb < 4; ++fb)
                    acc[fb] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][fb][s], zf[m][s], acc[fb], 0, 0, 0);
            }
        }
        // T^T (lane: atom i, features 16 fb + 4 q ..) -> the tile (chunk c of row r at position c ^ r) -> row-major lanes
#pragma unroll
        for (int fb = 0; fb < 4; ++fb) *reinterpret_cast<f32x4v*>(tl + i * 64 + (((4 * fb + q) ^ i) << 2)) = acc[fb];
        f32x4v o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = *reinterpret_cast<const f32x4v*>(tl + (4 * j + q) * 64 + ((i ^ (4 * j + q)) << 2));
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] += g0[j] + g1[j];
        for (int p = 2; p < np; p += 2) {                          // further neighbours, two rows in flight
            const bool two = p + 1 < np;
            if (p == 2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    nb0[j] = nb2[j];
                    nb1[j] = two ? nb3[j] : nb2[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    nb0[j] = (unsigned)adj_d[(unsigned)(lr[j] * deg + p)];
                    nb1[j] = two ? (unsigned)adj_d[(unsigned)(lr[j] * deg + p + 1)] : nb0[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                g0[j] = *reinterpret_cast<const f32x4v*>(gbuf + (size_t)min(nb0[j], nmax) * 64 + 4 * i);
                g1[j] = *reinterpret_cast<const f32x4v*>(gbuf + (size_t)min(nb1[j], nmax) * 64 + 4 * i);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) o[j] += two ? g0[j] + g1[j] : g0[j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // the tile's T has been read: the plane is free
        if (t + 4 < t_last) {
            issue_z(t + 4);
            load_nb(t + 4);
        }
        asm volatile("" ::: "memory");
        // (rows past the bucket's end repeat its last row: same values to the same address)
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4v*>(dx + (size_t)(bstart + lr[j]) * 64 + 4 * i) = o[j];
        asm volatile("" ::: "memory");
    }
}

// asymmetric adjacency only: dx[n] += G[a] for every neighbour n of a (float atomics); returns at once otherwise
__global__ void __launch_bounds__(256)
graphconv_bwd_dx_scatter_kernel(int n_atoms, const long long* __restrict__ deg_slice, GcParams P,
                                const float* __restrict__ gbuf, float* __restrict__ dx, const int* __restrict__ sym_flag) {
    if (*sym_flag != 0) return;
    const int l16 = threadIdx.x & 15;
    for (long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4); row < n_atoms; row += (long long)gridDim.x * 16) {
        int deg;
        long long lrow;
        find_bucket(deg_slice, row, deg, lrow);
        if (deg == 0) continue;
        const int* arow = P.adj[deg - 1] + lrow * deg;
        const float4 g = *reinterpret_cast<const float4*>(gbuf + row * 64 + 4 * l16);
        for (int n = 0; n < deg; ++n) {
            float* d = dx + (long long)min(max(arow[n], 0), n_atoms - 1) * 64 + 4 * l16;
            atomicAdd(d, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
        }
    }
}

// ================================================================================================ C ABI
static int g_num_cu = 0;
static hipError_t num_cu_cached() {
    if (g_num_cu == 0) {
        int dev = 0, cu = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        g_num_cu = cu > 0 ? cu : 1;
    }
    return hipSuccess;
}
static int fill_params(GcParams& P, const int32_t* const* adj, const float* const* w, const float* const* b) {
    for (int i = 0; i < 10; ++i) P.adj[i] = adj ? adj[i] : nullptr;
    for (int i = 0; i < 21; ++i) {
        P.w[i] = w ? w[i] : nullptr;
        P.b[i] = b ? b[i] : nullptr;
    }
    return 0;
}

// Host work plan of the 32-atom-tile kernels (see BwdPlan): buckets in descending degree, every workgroup about the
// same number of steps (a tile of degree d takes max(1, ceil(d / nbp)) gather rounds), at most slots_per_cu x #CU
// workgroups so that all are resident at once.  Returns the number of blocks, or -1 with the error set.
static int make_plan(const int64_t* dsh, int64_t n_atoms, int nbp, int slots_per_cu, BwdPlan& plan) {
    memset(&plan, 0, sizeof(plan));
    if (num_cu_cached() != hipSuccess) { jova_set_error("make_plan: cannot query the device"); return -1; }
    long long total = 0, tl[NB];
    int rounds[NB];
    for (int d = 0; d < NB; ++d) {
        const long long st = dsh[2 * d], sz = dsh[2 * d + 1];
        if (sz < 0 || st < 0 || st + sz > n_atoms) {
            jova_set_error("deg_slice_host[%d] = (%lld, %lld) out of range for %lld atoms", d, st, sz, (long long)n_atoms);
            return -1;
        }
        tl[d] = (sz + GT - 1) / GT;
        rounds[d] = d > nbp ? (d + nbp - 1) / nbp : 1;
        total += tl[d] * rounds[d];
        plan.start[d] = (int)st;
        plan.size[d] = (int)sz;
    }
    const long long slots = (long long)slots_per_cu * g_num_cu;
    long long target = total > slots ? (total + slots - 1) / slots : 1;      // steps per workgroup
    for (;; ++target) {
        long long blk = 0;
        for (int d = NB - 1; d >= 0; --d) {
            const long long tpw = target / rounds[d] > 1 ? target / rounds[d] : 1;
            plan.tpw[d] = (int)tpw;
            blk += (tl[d] + tpw - 1) / tpw;
        }
        if (blk <= slots || target > total) break;
    }
    int blk = 0;
    plan.first[NB] = 0;
    for (int d = NB - 1; d >= 0; --d) {
        blk += (int)((tl[d] + plan.tpw[d] - 1) / plan.tpw[d]);
        plan.first[d] = blk;
    }
    plan.use = 1;
    return blk > 0 ? blk : 1;
}

// Work split of the wave-private forward: workgroups per bucket in proportion to its cost (a tile of degree d costs
// one MFMA phase -- about half of one for d = 0 -- plus a share per extra gather round), <= num_wg workgroups in all.
static int make_wave_plan(const int64_t* dsh, int64_t n_atoms, int nbr, int num_wg, WavePlan& plan, bool bwd = false) {
    memset(&plan, 0, sizeof(plan));
    double cost[NB], total = 0.0;
    for (int d = 0; d < NB; ++d) {
        const long long st = dsh[2 * d], sz = dsh[2 * d + 1];
        if (sz < 0 || st < 0 || st + sz > n_atoms) {
            jova_set_error("deg_slice_host[%d] = (%lld, %lld) out of range for %lld atoms", d, st, sz, (long long)n_atoms);
            return -1;
        }
        plan.start[d] = (int)st;
        plan.size[d] = (int)sz;
        // cycles per tile relative to a degree-1 tile, measured with in-kernel stamps on the config-4 batch
        // (tools/prof_wave.py): the MFMA phase is the same for every degree >= 1, the gather rounds are not
        static const double rel3[NB] = {0.6, 1.0, 1.08, 1.18, 1.40, 1.46, 1.54, 1.78, 1.82, 1.90, 2.15};
        static const double rel2[NB] = {0.6, 1.0, 1.08, 1.35, 1.45, 1.70, 1.80, 2.05, 2.15, 2.40, 2.50};
        // backward (two neighbour planes, G + dW phases; tools/prof_wave_bwd.py)
        static const double relb[NB] = {0.45, 1.0, 1.08, 1.39, 1.47, 1.71, 1.77, 1.99, 2.06, 2.30, 2.36};
        // backward with waves by matrix (four neighbour planes: a second gather round only beyond degree 4)
        // (measured per tile of the dW_rel waves, which bound it: tools/prof_split_bwd.py)
        static const double rels[NB] = {0.65, 1.0, 1.12, 1.21, 1.24, 1.80, 2.07, 2.27, 2.48, 3.03, 3.27};
        cost[d] = (double)((sz + WT - 1) / WT) * (bwd ? (nbr >= 4 ? rels[d] : relb[d]) : nbr >= 3 ? rel3[d] : rel2[d]);
        total += cost[d];
    }
    if (total <= 0.0) return 0;
    int wgs[NB];
    for (double target = total / num_wg;; target *= 1.02) {
        int sum = 0;
        for (int d = 0; d < NB; ++d) {
            const long long tl = (plan.size[d] + WT - 1) / WT;
            long long w = cost[d] > 0.0 ? (long long)(cost[d] / target + 0.999) : 0;
            if (w > (tl + 7) / 8) w = (tl + 7) / 8;          // at least one tile per wave
            if (cost[d] > 0.0 && w < 1) w = 1;
            wgs[d] = (int)w;
            sum += wgs[d];
        }
        if (sum <= num_wg) break;
    }
    int blk = 0;
    for (int k = 0; k < NB; ++k) {                       // dispatch order: descending degree
        plan.first[k] = blk;
        blk += wgs[NB - 1 - k];
    }
    plan.first[NB] = blk;
    return blk;
}

extern "C" size_t jova_graphconv_workspace_floats(int f_in, int f_out) {
    const int kh = (f_in + 1) / 2, ncb = (f_out + 31) / 32, kg = (f_in + 7) / 8;
    const size_t a = (size_t)21 * kh * ncb * 64, b = (size_t)21 * ncb * kg * 256;
    const size_t c = (size_t)21 * 4 * ((kg + 1) / 2) * 256;                 // wave-private form (f_out = 64)
    return std::max(a, std::max(b, c));
}

extern "C" size_t jova_graphconv_bwd_scratch_floats(void) { return (size_t)21 * 64 * DWP_LD; }

extern "C" int jova_atoms_to_f32(const int64_t* atoms, int64_t n_atoms, int f, int ld, float* out, void* stream) {
    JOVA_CHECK_ARG(atoms && out && n_atoms >= 0 && f >= 1 && ld >= f, "atoms_to_f32: bad argument");
    if (n_atoms == 0) return JOVA_OK;
    const long long total = n_atoms * ld;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(atoms_to_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const long long*)atoms,
                       (long long)n_atoms, f, ld, out);
    JOVA_CHECK_LAUNCH("atoms_to_f32");
    return JOVA_OK;
}

extern "C" int jova_graphconv_fwd(const void* atoms, int atoms_is_i64, int64_t ld_atoms, int64_t n_atoms, int f_in, int f_out,
                                  const int64_t* deg_slice, const int32_t* const* adj, const float* const* weight,
                                  const float* const* bias, int relu, float* out, double* bn_stats,
                                  float* packed_ws, const int64_t* deg_slice_host, void* stream) {
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31), "graphconv_fwd: n_atoms %lld out of range", (long long)n_atoms);
    JOVA_CHECK_ARG(f_in >= 1 && f_in <= 128, "graphconv_fwd: f_in %d not in [1,128]", f_in);
    JOVA_CHECK_ARG(f_out >= 1, "graphconv_fwd: f_out %d", f_out);
    JOVA_CHECK_ARG(ld_atoms >= f_in && ld_atoms < (1 << 20), "graphconv_fwd: ld_atoms %lld < f_in", (long long)ld_atoms);
    // (an empty batch has no per-atom buffers: those pointers may be null when n_atoms == 0)
    JOVA_CHECK_ARG(deg_slice && adj && weight && bias && packed_ws && (n_atoms == 0 || (atoms && out)), "graphconv_fwd: null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (bn_stats) JOVA_ZERO(bn_stats, sizeof(double) * 2 * f_out, s);
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, weight, bias);
    const int kh = (f_in + 1) / 2, ncb = (f_out + 31) / 32;
    const int kg = (f_in + 7) / 8;
    // fast form: fp32 rows of 64 or 80 floats read whole by 16-byte LDS-DMA; columns f_in .. 8 kg - 1 of a row are
    // multiplied by zero weights, so they must exist (ld >= 8 kg) and hold finite values (the fp32 copy of int64
    // atoms is zero padded)
    // (the kernel's addressing is 24 x 24-bit multiplies and 32-bit element offsets)
    if (!atoms_is_i64 && (kg == 8 || kg == 10) && ld_atoms >= 8 * kg && (ld_atoms & 3) == 0 &&
        n_atoms < (1 << 24) && n_atoms * ld_atoms < (1ll << 31) && (((uintptr_t)atoms) & 15) == 0 &&
        !getenv("JOVA_GRAPHCONV_NO_DMA")) {
        // wave-private form: f_out = 64, bucket sizes known on the host, output rows 16-byte aligned
        static int wave_env = -1;
        if (wave_env < 0) wave_env = getenv("JOVA_FWD_OLD") ? 0 : 1;
        if (wave_env && f_out == 64 && deg_slice_host && (((uintptr_t)out) & 15) == 0) {
            JOVA_CHECK_HIP(num_cu_cached());
            const int nm = kg / 2;
            WavePlan wplan;
            const int nblk = make_wave_plan(deg_slice_host, n_atoms, kg == 8 ? 3 : 2, g_num_cu, wplan);
            if (nblk < 0) return JOVA_ERR_ARG;
            if (nblk == 0) return JOVA_OK;
            (void)nm;
            // (the weights go from the layer's matrices into LDS fragment order inside the kernel: no re-pack launch, the
            // workspace is not used by this form)
#define GC_WAVE(KG_, NPL_, ST_)                                                                                        \
    hipLaunchKernelGGL((graphconv_fwd_wave_kernel<KG_, NPL_, ST_>), dim3(nblk), dim3(512), 0, s, (const float*)atoms,  \
                       (int)n_atoms, (int)ld_atoms, f_in, P, relu, out, bn_stats, wplan)
            if (kg == 8) { if (bn_stats) GC_WAVE(8, 4, true); else GC_WAVE(8, 4, false); }
            else { if (bn_stats) GC_WAVE(10, 3, true); else GC_WAVE(10, 3, false); }
#undef GC_WAVE
            JOVA_CHECK_LAUNCH("graphconv_fwd_wave");
            return JOVA_OK;
        }
        const long long qtotal = 21ll * ncb * kg * 256;
        hipLaunchKernelGGL(graphconv_pack_dma_kernel, dim3(cdiv(qtotal, 256)), dim3(256), 0, s, P, f_in, f_out, kg, ncb,
                           packed_ws);
        JOVA_CHECK_LAUNCH("graphconv_pack_dma");
        const int tiles = cdiv(n_atoms, GT);
        int group = tiles > 2048 ? 4 : (tiles > 512 ? 2 : 1);
        if (getenv("JOVA_FWD_GROUP")) group = atoi(getenv("JOVA_FWD_GROUP"));
        dim3 grid(cdiv(tiles, group) + NB, cdiv(f_out, 64));
        static int npl_env = -1;
        if (npl_env < 0) {
            const char* e = getenv("JOVA_GRAPHCONV_PLANES");
            npl_env = e ? atoi(e) : 0;
        }
        const int npl = npl_env == 8 ? 8 : 4;
        // bucket sizes known on the host: workgroups of about equal length (a tile of degree d takes ceil(d / (npl - 1))
        // gather rounds), four per CU
        BwdPlan plan;
        memset(&plan, 0, sizeof(plan));
        if (deg_slice_host && !getenv("JOVA_FWD_NOPLAN")) {
            const int nblk = make_plan(deg_slice_host, n_atoms, npl - 1, getenv("JOVA_FWD_SLOTS") ? atoi(getenv("JOVA_FWD_SLOTS")) : (npl == 4 ? 4 : 2), plan);
            if (nblk < 0) return JOVA_ERR_ARG;
            grid.x = nblk;
        }
#define GC_DMA(KG_, NPL_)                                                                                              \
    hipLaunchKernelGGL((graphconv_fwd_dma_kernel<KG_, NPL_>), grid, dim3(128), 0, s, (const float*)atoms,              \
                       (int)n_atoms, (int)ld_atoms, f_out, (const long long*)deg_slice, P, packed_ws, ncb, relu, out,  \
                       bn_stats, group, plan)
        if (kg == 8 && npl == 4) GC_DMA(8, 4);
        else if (kg == 8) GC_DMA(8, 8);
        else if (npl == 4) GC_DMA(10, 4);
        else GC_DMA(10, 8);
#undef GC_DMA
        JOVA_CHECK_LAUNCH("graphconv_fwd_dma");
        return JOVA_OK;
    }
    const long long wtotal = 21ll * kh * ncb * 64;
    hipLaunchKernelGGL(graphconv_pack_kernel, dim3(cdiv(wtotal, 256)), dim3(256), 0, s, P, f_in, f_out, kh, ncb,
                       packed_ws);
    JOVA_CHECK_LAUNCH("graphconv_pack");
    const int tiles = cdiv(n_atoms, TM);
    const int group = tiles > 4096 ? cdiv(tiles, 2048) : 1;
    dim3 grid(cdiv(tiles, group) + NB, cdiv(f_out, 64));
    const size_t lds = sizeof(float) * 2 * TM * (2 * kh + 1);
    if (atoms_is_i64)
        hipLaunchKernelGGL(graphconv_fwd_kernel<long long>, grid, dim3(256), lds, s, (const long long*)atoms,
                           (int)n_atoms, f_in, (int)ld_atoms, f_out, (const long long*)deg_slice, P, packed_ws, kh, ncb,
                           relu, out, bn_stats, group);
    else
        hipLaunchKernelGGL(graphconv_fwd_kernel<float>, grid, dim3(256), lds, s, (const float*)atoms, (int)n_atoms,
                           f_in, (int)ld_atoms, f_out, (const long long*)deg_slice, P, packed_ws, kh, ncb, relu, out,
                           bn_stats, group);
    JOVA_CHECK_LAUNCH("graphconv_fwd");
    return JOVA_OK;
}

extern "C" int jova_graphconv_bwd(const void* atoms, int atoms_is_i64, int64_t ld_atoms, int64_t n_atoms, int f_in,
                                  int f_out, const int64_t* deg_slice, const int32_t* const* adj,
                                  const float* const* weight, const float* dy, const float* y_or_null,
                                  float* const* dweight, float* const* dbias, float* dx_or_null, float* gbuf_or_null,
                                  const int* sym_flag, const int64_t* deg_slice_host, float* dw_scratch_or_null,
                                  const int32_t* block_order_or_null, void* stream) {
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31), "graphconv_bwd: n_atoms out of range");
    JOVA_CHECK_ARG(f_in >= 1 && f_in <= 128, "graphconv_bwd: f_in %d not in [1,128]", f_in);
    JOVA_CHECK_ARG(f_out >= 1 && f_out <= 64, "graphconv_bwd: f_out %d not in [1,64]", f_out);
    JOVA_CHECK_ARG(ld_atoms >= f_in && ld_atoms < (1 << 20), "graphconv_bwd: ld_atoms %lld < f_in", (long long)ld_atoms);
    JOVA_CHECK_ARG(deg_slice && adj && dweight && dbias && (n_atoms == 0 || (atoms && dy)), "graphconv_bwd: null pointer");
    JOVA_CHECK_ARG(!dx_or_null || (gbuf_or_null && weight && !atoms_is_i64), "graphconv_bwd: dx needs gbuf, weights, f32 atoms");
    if (n_atoms == 0) return JOVA_OK;
    hipStream_t s = (hipStream_t)stream;
    GcParams P;
    fill_params(P, adj, weight, nullptr);
    GcGrads G;
    for (int i = 0; i < 21; ++i) { G.dw[i] = dweight[i]; G.db[i] = dbias[i]; }
    const int kg = (f_in + 7) / 8;
    const bool al16 = ((((uintptr_t)atoms) | ((uintptr_t)dy) | ((uintptr_t)y_or_null) | ((uintptr_t)dx_or_null) |
                        ((uintptr_t)gbuf_or_null)) & 15) == 0;
    // fast form (see graphconv_bwd_dma_kernel): whole 256-byte rows by LDS-DMA; pad columns f_in .. 8 kg - 1 must exist
    // and hold finite values (their products are not stored)
    if (!atoms_is_i64 && f_out == 64 && (kg == 8 || kg == 10) && ld_atoms >= 8 * kg && (ld_atoms & 3) == 0 && al16 &&
        n_atoms < (1 << 24) && n_atoms * ld_atoms < (1ll << 31) &&
        (!dx_or_null || (kg == 8 && (f_in & 3) == 0)) && !getenv("JOVA_GRAPHCONV_NO_DMA")) {
        // weight rows that are not whole 128-byte lines: accumulate in the aligned scratch image, fold afterwards
        float* dwp = (dw_scratch_or_null && (f_in & 31) != 0 && !getenv("JOVA_BWD_NO_DWP")) ? dw_scratch_or_null : nullptr;
        if (dwp) JOVA_ZERO(dwp, sizeof(float) * 21 * 64 * DWP_LD, s);
        // wave-private form (see graphconv_bwd_wave_kernel): dZ handed in, host copy of the bucket table at hand
        // 80-float rows, dW only: waves specialised by matrix (graphconv_bwd_split_kernel)
        if (kg == 10 && !dx_or_null && !y_or_null && deg_slice_host && n_atoms * ld_atoms < (1ll << 30) &&
            !getenv("JOVA_BWD_OLD") && !getenv("JOVA_BWD_NOPLAN") && !getenv("JOVA_BWD_DB")) {
            if (num_cu_cached() != hipSuccess) { jova_set_error("graphconv_bwd: cannot query the device"); return JOVA_ERR_HIP; }
            WavePlan wplan;
            const int nblk = make_wave_plan(deg_slice_host, n_atoms, 4, g_num_cu, wplan, true);
            if (nblk < 0) return JOVA_ERR_ARG;
            if (nblk > 0) {
                hipLaunchKernelGGL((graphconv_bwd_split_kernel<10>), dim3(nblk), dim3(512), 0, s, (const float*)atoms,
                                   (int)n_atoms, (int)ld_atoms, f_in, P, dy, G, dwp, wplan);
                JOVA_CHECK_LAUNCH("graphconv_bwd_split");
            }
            if (dwp) {
                hipLaunchKernelGGL(graphconv_dw_fold_kernel, dim3(cdiv(21 * 64 * f_in, 256)), dim3(256), 0, s, dwp, G, f_in);
                JOVA_CHECK_LAUNCH("graphconv_dw_fold");
            }
            return JOVA_OK;
        }
        if (kg == 8 && !y_or_null && deg_slice_host && (!dx_or_null || f_in == 64) && n_atoms * ld_atoms < (1ll << 30) &&
            !getenv("JOVA_BWD_OLD") && !getenv("JOVA_BWD_NOPLAN") && !getenv("JOVA_BWD_DB")) {
            if (num_cu_cached() != hipSuccess) { jova_set_error("graphconv_bwd: cannot query the device"); return JOVA_ERR_HIP; }
            WavePlan wplan;
            const int nblk = make_wave_plan(deg_slice_host, n_atoms, 2, g_num_cu, wplan, true);
            if (nblk < 0) return JOVA_ERR_ARG;
            if (nblk > 0) {
#define GC_BWD_WAVE(KG_, DX_, NBR_)                                                                                       \
    hipLaunchKernelGGL((graphconv_bwd_wave_kernel<KG_, DX_, NBR_>), dim3(nblk), dim3(512), 0, s, (const float*)atoms,        \
                       (int)n_atoms, (int)ld_atoms, f_in, P, dy, G, gbuf_or_null, dx_or_null, dwp, wplan)
                if (dx_or_null) GC_BWD_WAVE(8, true, 2);
                else GC_BWD_WAVE(8, false, 2);
#undef GC_BWD_WAVE
                JOVA_CHECK_LAUNCH("graphconv_bwd_wave");
            }
            if (dwp) {
                hipLaunchKernelGGL(graphconv_dw_fold_kernel, dim3(cdiv(21 * 64 * f_in, 256)), dim3(256), 0, s, dwp, G, f_in);
                JOVA_CHECK_LAUNCH("graphconv_dw_fold");
            }
            if (dx_or_null) {
                JOVA_CHECK_ARG(sym_flag, "graphconv_bwd: sym_flag is null");
                // many small workgroups (4 waves, `tpw` tiles per wave, one bucket each), descending degree
                WavePlan dplan;
                memset(&dplan, 0, sizeof(dplan));
                long long total_tiles = 0;
                for (int d = 0; d < NB; ++d) total_tiles += (deg_slice_host[2 * d + 1] + WT - 1) / WT;
                // one tile per wave up to 2^16 workgroups (measured on the config-4 batch: 42.7 / 46.1 / 59.1 / 84.9 us
                // with 1 / 2 / 4 / 8 tiles per wave -- the gather latency of a tile is hidden by other waves, not by
                // the wave's own next tile)
                const int tpw = (int)std::max<long long>(1, (total_tiles + 4 * 65536 - 1) / (4 * 65536));
                int blk = 0;
                for (int k = 0; k < NB; ++k) {
                    const int d = NB - 1 - k;
                    dplan.start[d] = (int)deg_slice_host[2 * d];
                    dplan.size[d] = (int)deg_slice_host[2 * d + 1];
                    dplan.first[k] = blk;
                    blk += (int)(((dplan.size[d] + WT - 1) / WT + 4 * tpw - 1) / (4 * tpw));
                }
                dplan.first[NB] = blk;
                if (blk > 0) {
                    hipLaunchKernelGGL(graphconv_bwd_dx_wave_kernel, dim3(blk), dim3(256), 0, s, dy, (int)n_atoms, P,
                                       gbuf_or_null, dx_or_null, sym_flag, dplan, tpw);
                    JOVA_CHECK_LAUNCH("graphconv_bwd_dx_wave");
                }
                hipLaunchKernelGGL(graphconv_bwd_dx_scatter_kernel, dim3(2 * g_num_cu), dim3(256), 0, s, (int)n_atoms,
                                   (const long long*)deg_slice, P, gbuf_or_null, dx_or_null, sym_flag);
                JOVA_CHECK_LAUNCH("graphconv_bwd_dx_scatter");
            }
            return JOVA_OK;
        }
        const int tiles = cdiv(n_atoms, GT);
        int group = tiles > 2048 ? cdiv(tiles, 2000) : 1;
        if (getenv("JOVA_BWD_GROUP")) group = atoi(getenv("JOVA_BWD_GROUP"));
        dim3 grid(cdiv(tiles, group) + NB);
        // Two LDS buffers (JOVA_BWD_DB=1, only without the Y plane) measured no faster than one on the config-4 batch
        // once the work plan was in (119 vs 117 us at 64 -> 64; 167 vs 128 us at 75 -> 64, where two buffers leave room
        // for two neighbour planes only): with two workgroups per CU one computes while the other gathers.
        const bool nodb = getenv("JOVA_BWD_DB") == nullptr;
        const bool db = !y_or_null && !nodb;
        const int nbp = db ? (kg == 8 ? 3 : 2) : 4;
        BwdPlan plan;
        memset(&plan, 0, sizeof(plan));
        if (deg_slice_host && !getenv("JOVA_BWD_NOPLAN")) {
            const int nblk = make_plan(deg_slice_host, n_atoms, nbp, 2, plan);
            if (nblk < 0) return JOVA_ERR_ARG;
            grid = dim3(nblk);
        }
#define GC_BWD_DMA(KG_, DX_, M_, NBP_, DB_)                                                                            \
    hipLaunchKernelGGL((graphconv_bwd_dma_kernel<KG_, DX_, M_, NBP_, DB_>), grid, dim3(256), 0, s, (const float*)atoms,\
                       (int)n_atoms, (int)ld_atoms, f_in, (const long long*)deg_slice, P, dy, y_or_null, G,            \
                       gbuf_or_null, dx_or_null, group, plan, dwp)
        // dZ handed in (the consumer BatchNorm applied the ReLU mask): two LDS buffers of [own, dZ, 3 (2) neighbour
        // planes] fit two workgroups per CU; with the Y plane only the single-buffer form does
        if (dx_or_null) {
            if (y_or_null || nodb) { if (y_or_null) GC_BWD_DMA(8, true, true, 4, false); else GC_BWD_DMA(8, true, false, 4, false); }
            else GC_BWD_DMA(8, true, false, 3, true);
        } else if (kg == 8) {
            if (y_or_null || nodb) { if (y_or_null) GC_BWD_DMA(8, false, true, 4, false); else GC_BWD_DMA(8, false, false, 4, false); }
            else GC_BWD_DMA(8, false, false, 3, true);
        } else {
            if (y_or_null || nodb) { if (y_or_null) GC_BWD_DMA(10, false, true, 4, false); else GC_BWD_DMA(10, false, false, 4, false); }
            else GC_BWD_DMA(10, false, false, 2, true);
        }
#undef GC_BWD_DMA
        JOVA_CHECK_LAUNCH("graphconv_bwd_dma");
        if (dwp) {
            hipLaunchKernelGGL(graphconv_dw_fold_kernel, dim3(cdiv(21 * 64 * f_in, 256)), dim3(256), 0, s, dwp, G, f_in);
            JOVA_CHECK_LAUNCH("graphconv_dw_fold");
        }
        if (dx_or_null) {
            JOVA_CHECK_ARG(sym_flag, "graphconv_bwd: sym_flag is null");
            hipLaunchKernelGGL(graphconv_bwd_dx_v4_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, s, (int)n_atoms, f_in,
                               (const long long*)deg_slice, P, gbuf_or_null, dx_or_null, sym_flag, block_order_or_null);
            JOVA_CHECK_LAUNCH("graphconv_bwd_dx_v4");
        }
        return JOVA_OK;
    }
    const int NT = (f_in + 31) / 32, KS = 32 * NT + 1;
    const size_t lds = sizeof(float) * (TM * 65 + 2 * TM * KS);
    const int tiles = cdiv(n_atoms, TM);
    const int group = tiles > 1024 ? cdiv(tiles, 768) : 1;
    dim3 grid(cdiv(tiles, group) + NB);
#define GC_BWD(T, DX, G_, D_) hipLaunchKernelGGL((graphconv_bwd_kernel<T, DX>), grid, dim3(256), lds, s, (const T*)atoms, \
                                  (int)n_atoms, f_in, (int)ld_atoms, f_out, (const long long*)deg_slice, P, dy, y_or_null, G, G_, D_, group)
    if (atoms_is_i64) GC_BWD(long long, false, nullptr, nullptr);
    else if (!dx_or_null) GC_BWD(float, false, nullptr, nullptr);
    else GC_BWD(float, true, gbuf_or_null, dx_or_null);
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
    hipLaunchKernelGGL(set_int_kernel, dim3(1), dim3(1), 0, s, flag, 1);      // (no host memory in the stream: capturable)
    JOVA_CHECK_LAUNCH("adj_symmetry_init");
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    hipLaunchKernelGGL(adj_symmetry_kernel, dim3(cdiv(n_atoms, 256)), dim3(256), 0, s, (int)n_atoms,
                       (const long long*)deg_slice, P, flag);
    JOVA_CHECK_LAUNCH("adj_symmetry");
    return JOVA_OK;
}

extern "C" int jova_adj_reverse_pos(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj,
                                    uint8_t* rev_pos, void* stream) {
    JOVA_CHECK_ARG(deg_slice && adj && rev_pos, "adj_reverse_pos: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31), "adj_reverse_pos: n_atoms out of range");
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    hipLaunchKernelGGL(adj_reverse_pos_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, (hipStream_t)stream, (int)n_atoms,
                       (const long long*)deg_slice, P, rev_pos);
    JOVA_CHECK_LAUNCH("adj_reverse_pos");
    return JOVA_OK;
}

// The row table of the gather kernels (see row_table_kernel): 8 ints per slot, 16 * ceil(n_atoms/16) slots, in the
// launch order block_order gives (identity when null).  rev_pos may be null (the packed positions are then 0xFF: the
// table still serves GraphPool's forward).
extern "C" int jova_row_table(int64_t n_atoms, const int64_t* deg_slice, const int32_t* const* adj,
                              const uint8_t* rev_pos_or_null, const int32_t* block_order_or_null, int32_t* table,
                              void* stream) {
    JOVA_CHECK_ARG(deg_slice && adj && (n_atoms == 0 || table), "row_table: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 27), "row_table: n_atoms out of range");
    JOVA_CHECK_ARG((((uintptr_t)table) & 15) == 0, "row_table: table must be 16-byte aligned");
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    const int n_blocks = (int)cdiv(n_atoms, 16), n_slots = n_blocks * 16;
    hipLaunchKernelGGL(row_table_kernel, dim3(cdiv(n_slots, 256)), dim3(256), 0, (hipStream_t)stream, (int)n_atoms,
                       n_slots, (const long long*)deg_slice, P, rev_pos_or_null, block_order_or_null, n_blocks,
                       (int4*)table);
    JOVA_CHECK_LAUNCH("row_table");
    return JOVA_OK;
}

// Host-side: the XCD-aware order of the 16-row blocks (see ordered_block).  order_host receives ceil(n_atoms/16) ints,
// a permutation of the block numbers.
extern "C" int jova_block_order(const int64_t* deg_slice_host, int64_t n_atoms, int32_t* order_host) {
    JOVA_CHECK_ARG(deg_slice_host && order_host && n_atoms >= 0 && n_atoms < (1ll << 31), "block_order: bad argument");
    const int n = (int)cdiv(n_atoms, 16);
    if (n == 0) return JOVA_OK;
    std::vector<std::pair<double, int>> key((size_t)n);
    for (int t = 0; t < n; ++t) {
        const long long row0 = 16ll * t;
        double rel = 0.0;
        for (int d = 0; d < NB; ++d) {
            const long long st = deg_slice_host[2 * d], sz = deg_slice_host[2 * d + 1];
            if (row0 >= st && row0 < st + sz) rel = (double)(row0 - st) / (double)sz;
        }
        key[(size_t)t] = std::make_pair(rel, t);
    }
    std::stable_sort(key.begin(), key.end());
    const int q = n / 8, r = n % 8;
    int off = 0;
    for (int k = 0; k < 8; ++k) {                                  // XCD k: the k-th eighth of the sorted list
        const int cnt = q + (k < r ? 1 : 0);
        for (int loc = 0; loc < cnt; ++loc) order_host[loc * 8 + k] = key[(size_t)(off + loc)].second;
        off += cnt;
    }
    return JOVA_OK;
}

extern "C" int jova_graphpool_fwd(const float* x, int64_t n_atoms, int f, const int64_t* deg_slice,
                                  const int32_t* const* adj, float* out, uint8_t* argmax,
                                  const int32_t* block_order_or_null, const int32_t* row_table_or_null, void* stream) {
    JOVA_CHECK_ARG(deg_slice && adj && (n_atoms == 0 || (x && out && argmax)), "graphpool_fwd: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31) && f >= 1, "graphpool_fwd: bad shape");
    if (n_atoms == 0) return JOVA_OK;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    const bool vec = (f & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)argmax)) & 15) == 0;
    if (vec && row_table_or_null && (((uintptr_t)row_table_or_null) & 15) == 0)
        hipLaunchKernelGGL(graphpool_fwd_t_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, (hipStream_t)stream, x,
                           (int)n_atoms, f, (const long long*)deg_slice, P, out, argmax,
                           (const int4*)row_table_or_null);
    else if (vec)
        hipLaunchKernelGGL(graphpool_fwd_v4_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, (hipStream_t)stream, x,
                           (int)n_atoms, f, (const long long*)deg_slice, P, out, argmax, block_order_or_null);
    else
        hipLaunchKernelGGL(graphpool_fwd_kernel, dim3(cdiv(n_atoms, 4)), dim3(256), 0, (hipStream_t)stream, x,
                           (int)n_atoms, f, (const long long*)deg_slice, P, out, argmax);
    JOVA_CHECK_LAUNCH("graphpool_fwd");
    return JOVA_OK;
}

extern "C" int jova_graphpool_bwd(const float* dy, const uint8_t* argmax, int64_t n_atoms, int f,
                                  const int64_t* deg_slice, const int32_t* const* adj, float* dx, const int* sym_flag,
                                  int sym_flag_host_hint, const uint8_t* rev_pos_or_null,
                                  const int32_t* block_order_or_null, const int32_t* row_table_or_null, void* stream) {
    JOVA_CHECK_ARG(deg_slice && adj && (n_atoms == 0 || (dy && argmax && dx)), "graphpool_bwd: null pointer");
    JOVA_CHECK_ARG(n_atoms >= 0 && n_atoms < (1ll << 31) && f >= 1, "graphpool_bwd: bad shape");
    if (n_atoms == 0) return JOVA_OK;
    hipStream_t s = (hipStream_t)stream;
    GcParams P;
    fill_params(P, adj, nullptr, nullptr);
    // the caller may vouch for a symmetric adjacency (it read the flag once per batch): 16-lanes-per-row gather form
    if (sym_flag_host_hint == 1 && (f & 3) == 0 && ((((uintptr_t)dy) | ((uintptr_t)dx) | ((uintptr_t)argmax)) & 15) == 0) {
        if (rev_pos_or_null && row_table_or_null && (((uintptr_t)row_table_or_null) & 15) == 0) {
            hipLaunchKernelGGL(graphpool_bwd_t_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, s, dy, argmax,
                               (int)n_atoms, f, (const long long*)deg_slice, P, rev_pos_or_null, dx,
                               (const int4*)row_table_or_null);
            JOVA_CHECK_LAUNCH("graphpool_bwd_t");
            return JOVA_OK;
        }
        if (rev_pos_or_null) {
            hipLaunchKernelGGL(graphpool_bwd_v5_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, s, dy, argmax,
                               (int)n_atoms, f, (const long long*)deg_slice, P, rev_pos_or_null, dx,
                               block_order_or_null);
            JOVA_CHECK_LAUNCH("graphpool_bwd_v5");
            return JOVA_OK;
        }
        hipLaunchKernelGGL(graphpool_bwd_v4_kernel, dim3(cdiv(n_atoms, 16)), dim3(256), 0, s, dy, argmax,
                           (int)n_atoms, f, (const long long*)deg_slice, P, dx);
        JOVA_CHECK_LAUNCH("graphpool_bwd_v4");
        return JOVA_OK;
    }
    // the scatter path accumulates into dx; zeroing is harmless for the gather path (it overwrites)
    JOVA_ZERO(dx, sizeof(float) * (size_t)n_atoms * f, s);
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
    if ((g & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out)) & 15) == 0 && (long long)a_max * batch < (1ll << 32))
        hipLaunchKernelGGL(gather2d_fwd_v4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, x,
                           mol_off, a_max, batch, g / 4, out);
    else
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
    if ((g & 3) == 0 && ((((uintptr_t)dout) | ((uintptr_t)dx)) & 15) == 0 && (long long)a_max * batch < (1ll << 32))
        hipLaunchKernelGGL(gather2d_bwd_v4_kernel, dim3(cdiv(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, dout,
                           mol_off, a_max, batch, g / 4, dx);
    else
        hipLaunchKernelGGL(gather2d_bwd_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, dout, mol_off,
                           a_max, batch, g, dx);
    JOVA_CHECK_LAUNCH("gather2d_bwd");
    return JOVA_OK;
}

extern "C" size_t jova_nonsat_workspace_bytes(int64_t n) {
    return 64 + sizeof(double) * 2 * NS_MAX_BLOCKS + sizeof(float) * 2 * (size_t)(n > 0 ? n : 0);
}

static unsigned long long g_ns_spin_limit = 200000000ull;      // ticks of the 100 MHz wall clock: 2 s
extern "C" void jova_nonsat_set_spin_limit(uint64_t ticks) { g_ns_spin_limit = ticks ? ticks : 200000000ull; }

// Workgroups the in-kernel barrier may use: one per CU, and never more than the occupancy query says fit at once.
template <int VEC>
static int nonsat_block_cap(int* cap) {
    static int cached = 0;
    if (cached == 0) {
        int per_cu = 0;
        JOVA_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, nonsat_persistent_kernel<VEC, (VEC == 4 ? NS_RESIDENT : 0)>,
                                                                     NS_THREADS, 0));
        if (per_cu < 1) {
            jova_set_error("nonsat_fwd: the persistent kernel does not fit a CU (occupancy query returned %d)", per_cu);
            return JOVA_ERR_LIMIT;
        }
        cached = g_num_cu < NS_MAX_BLOCKS ? g_num_cu : NS_MAX_BLOCKS;
    }
    *cap = cached;
    return JOVA_OK;
}

__global__ void nonsat_init_kernel(int* __restrict__ ws_words, int n_words, int* __restrict__ n_eval) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_words) ws_words[e] = 0;
    if (e < 3) n_eval[e] = 0;
}

static int nonsat_launch(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac, int* n_eval_dev,
                         void* workspace, hipStream_t s) {
    JOVA_CHECK_HIP(num_cu_cached());
    // workspace: ctrl[16 ints] | partial[2][NS_MAX_BLOCKS] doubles | ya[n] | yb[n]
    int* ctrl = (int*)workspace;
    double* partial = (double*)((char*)workspace + 64);
    float* ya = (float*)((char*)workspace + 64 + sizeof(double) * 2 * NS_MAX_BLOCKS);
    float* yb = ya + n;
    // control words, partials and the result words start at zero: ONE small kernel (two memsets before; as memset nodes
    // of a captured hipGraph they did not take effect on every replay -- tests/test_gpu_ops.py, capture test)
    hipLaunchKernelGGL(nonsat_init_kernel, dim3(cdiv(16 + 4 * NS_MAX_BLOCKS, 256)), dim3(256), 0, s, (int*)workspace,
                       16 + 4 * NS_MAX_BLOCKS, n_eval_dev);
    JOVA_CHECK_LAUNCH("nonsat_init");
    if (n == 0) return JOVA_OK;
    // one workgroup per CU at most, capped by the occupancy query: every workgroup is resident at once, which the
    // in-kernel barrier needs (a plain launch: the call stays capturable into a hipGraph)
    const bool vec4 = (n & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)out) | ((uintptr_t)dfac) | ((uintptr_t)ya)) & 15) == 0;
    const long long per_block = (long long)NS_THREADS * (vec4 ? 4 : 1);
    int blocks = (int)((n + per_block - 1) / per_block);
    int cap = 0;
    const int rc = vec4 ? nonsat_block_cap<4>(&cap) : nonsat_block_cap<1>(&cap);
    if (rc != JOVA_OK) return rc;
    if (blocks > cap) blocks = cap;
    // up to NS_RESIDENT 16-byte pieces per thread: the iterates never leave the registers (the Davis batch: 1 M elements)
    static int res_env = -1;
    if (res_env < 0) res_env = getenv("JOVA_NONSAT_NO_RESIDENT") ? 0 : 1;
    if (vec4 && res_env && n <= (long long)blocks * per_block * NS_RESIDENT)
        hipLaunchKernelGGL((nonsat_persistent_kernel<4, NS_RESIDENT>), dim3(blocks), dim3(NS_THREADS), 0, s, x, ya, yb,
                           (long long)n, ep, max_iter, partial, ctrl, out, dfac, n_eval_dev, g_ns_spin_limit);
    else if (vec4)
        hipLaunchKernelGGL((nonsat_persistent_kernel<4, 0>), dim3(blocks), dim3(NS_THREADS), 0, s, x, ya, yb, (long long)n, ep,
                           max_iter, partial, ctrl, out, dfac, n_eval_dev, g_ns_spin_limit);
    else
        hipLaunchKernelGGL((nonsat_persistent_kernel<1, 0>), dim3(blocks), dim3(NS_THREADS), 0, s, x, ya, yb, (long long)n, ep,
                           max_iter, partial, ctrl, out, dfac, n_eval_dev, g_ns_spin_limit);
    JOVA_CHECK_LAUNCH("nonsat_persistent");
    return JOVA_OK;
}

extern "C" int jova_nonsat_fwd(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac,
                               int* n_eval_dev, void* workspace, void* stream) {
    JOVA_CHECK_ARG(x && out && dfac && n_eval_dev && workspace && n >= 0, "nonsat_fwd: bad argument");
    JOVA_CHECK_ARG(max_iter >= 0 && max_iter <= 100000, "nonsat_fwd: max_iter %d", max_iter);
    return nonsat_launch(x, n, ep, max_iter, out, dfac, n_eval_dev, workspace, (hipStream_t)stream);
}

// ---- allocation-free variant (callers that cannot call hipMalloc: hipGraph capture, serving loops, harnesses that
// forbid it).  Same kernels as jova_nonsat_fwd; the workspace (control words, per-workgroup partials, the two iterate
// buffers) lives in a file-scope device array instead of caller memory.  Every GPU holds its own copy of the array at
// its own address: that address is looked up once per device with hipGetSymbolAddress and used from then on (the
// array's NAME is only a host-side stand-in and never reaches a kernel or a memset).
#define NS_NOALLOC_MAX_N (1ll << 21)              // elements; 2 x 8 MiB of iterates + 8.1 KiB of control per GPU
#define NS_NOALLOC_BYTES (64 + sizeof(double) * 2 * NS_MAX_BLOCKS + sizeof(float) * 2 * (size_t)NS_NOALLOC_MAX_N)
static __device__ __align__(256) unsigned char scratch[NS_NOALLOC_BYTES];
#define JOVA_MAX_DEVICES 64
static void* g_scratch_ptr[JOVA_MAX_DEVICES];
static std::mutex g_scratch_mu;

extern "C" int64_t jova_nonsat_noalloc_max_n(void) { return (int64_t)NS_NOALLOC_MAX_N; }

extern "C" int jova_nonsat_fwd_noalloc(const float* x, int64_t n, float ep, int max_iter, float* out, float* dfac,
                                       int* n_eval_dev, void* stream) {
    JOVA_CHECK_ARG(x && out && dfac && n_eval_dev && n >= 0, "nonsat_fwd_noalloc: bad argument");
    JOVA_CHECK_ARG(max_iter >= 0 && max_iter <= 100000, "nonsat_fwd_noalloc: max_iter %d", max_iter);
    if (n > NS_NOALLOC_MAX_N) {
        jova_set_error("nonsat_fwd_noalloc: %lld elements exceed the built-in scratch (%lld); use jova_nonsat_fwd with a "
                       "caller workspace", (long long)n, (long long)NS_NOALLOC_MAX_N);
        return JOVA_ERR_LIMIT;
    }
    // The scratch array is per GPU: it must be the copy of the GPU the work runs on, i.e. the device of `stream` and of
    // the buffers -- not whatever device the calling thread has current.  A null stream runs on the current device; any
    // other stream names its own.  The output buffer must live there too; a mismatch is an error, not a launch on GPU A
    // with GPU B's scratch address.
    int cur = 0, dev = 0;
    JOVA_CHECK_HIP(hipGetDevice(&cur));
    dev = cur;
    if (stream) {
        hipDevice_t sd = 0;
        JOVA_CHECK_HIP(hipStreamGetDevice((hipStream_t)stream, &sd));
        dev = (int)sd;
    }
    JOVA_CHECK_ARG(dev >= 0 && dev < JOVA_MAX_DEVICES, "nonsat_fwd_noalloc: device %d out of range", dev);
    {
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, out) == hipSuccess && at.type == hipMemoryTypeDevice && at.device != dev) {
            jova_set_error("nonsat_fwd_noalloc: `out` lives on GPU %d but the stream belongs to GPU %d", at.device, dev);
            return JOVA_ERR_ARG;
        }
        (void)hipGetLastError();
    }
    JOVA_CHECK_ARG(dev == cur, "nonsat_fwd_noalloc: the stream belongs to GPU %d but GPU %d is current: make the stream's "
                               "device current before the call", dev, cur);
    void* ws = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        if (!g_scratch_ptr[dev]) {               // first call on this device, with this device current (checked above)
            void* p = nullptr;
            JOVA_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            JOVA_CHECK_ARG(p != nullptr, "nonsat_fwd_noalloc: hipGetSymbolAddress returned null");
            g_scratch_ptr[dev] = p;
        }
        ws = g_scratch_ptr[dev];
    }
    return nonsat_launch(x, n, ep, max_iter, out, dfac, n_eval_dev, ws, (hipStream_t)stream);
}

extern "C" int jova_mul(const float* a, const float* b, int64_t n, float* out, void* stream) {
    JOVA_CHECK_ARG(a && b && out && n >= 0, "mul: bad argument");
    if (n == 0) return JOVA_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(mul_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n, out);
    JOVA_CHECK_LAUNCH("mul");
    return JOVA_OK;
}
