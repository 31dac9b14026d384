// fp32 MFMA GEMM for the dense contractions of the DTI hot path (view projections, PCNN, QKV, attention score / PV
// products, out-proj, FFN, FC head; forward, data-gradient and weight-gradient forms), gfx950.
//
//   C[z][m][n] = epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) )
//   A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; one of each stride pair must be 1
//   (K-contiguous "x[m][k]" / "W[n][k]" operands, or row-contiguous operands for the transposed products).
//   epilogue: + bias[n], + R[m*ldr + n] (residual), ReLU; or split-K partial sums added with float atomics.
//   z = z1 * nb2 + z2 selects the batch with two independent strides (needed for the [S,B,3L] QKV layout).
//
// v_mfma_f32_32x32x2_f32 is exact fp32 (k-ordered fma chain) at 64 FLOP/clk/SIMD, 64 cycles per instruction, so the
// kernel is MFMA-issue bound as long as the staging is hidden: 128x128x32 tiles, 4 waves as 2x2 of 64x64, register
// prefetch of the next k-tile under the current tile's 64 MFMAs, several workgroups per CU for latency hiding.
#include "common.h"
#include "jova_hip.h"
#include <stdlib.h>
#include <type_traits>

#define BK 32
#define LDS_KC (BK + 1)      // [row][k] image, odd stride: conflict-free ds_read_b32 down a column
#ifndef GEMM_WGS
#define GEMM_WGS 3           // workgroups per CU the register budget is held to
#endif

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* R;
    int M, N, K;
    long long sam, sak, sbk, sbn;
    long long ldc, ldr;
    int nb2;
    long long sA1, sA2, sB1, sB2, sC1, sC2;
    float alpha;
    int relu;
    int splitk;
    int vecA, vecB;
    float* csum;          // aligned form, row-contiguous A only: csum[m] += sum_k A[m][k] (bias gradient of a dW product)
};

// Branch-free guarded loads: out-of-range elements read the (always valid) tensor base, so every global load is
// unconditional and all of a tile's loads are in flight together (a branch per element would put an
// s_waitcnt vmcnt(0) behind each load: guide section 5, "Three .s-level traps" (c)).
// A staged quad: raw loaded data + how many of its 4 elements are inside the operand.  The zeroing select is applied
// when the quad is written to LDS (one iteration later), never right after the load, so the loads stay in flight
// under the MFMAs of the current tile.
struct Quad {
    float4 v;
    int nvalid;
};

template <bool VEC>
__device__ __forceinline__ Quad load4(const float* __restrict__ base, long long off, int nvalid) {
    Quad q;
    q.nvalid = nvalid;
    if (VEC) {      // VEC kernels are only launched when every quad that is partly inside the operand is readable as
                    // a whole (a multiple-of-4 extent, or rows padded to one: the host checks); masked() drops the rest
        q.v = *reinterpret_cast<const float4*>(base + (nvalid > 0 ? off : 0));
    } else {
        q.v.x = base[nvalid > 0 ? off : 0];
        q.v.y = base[nvalid > 1 ? off + 1 : 0];
        q.v.z = base[nvalid > 2 ? off + 2 : 0];
        q.v.w = base[nvalid > 3 ? off + 3 : 0];
    }
    return q;
}

__device__ __forceinline__ float4 masked(const Quad& q) {
    float4 r;
    r.x = q.nvalid > 0 ? q.v.x : 0.f;
    r.y = q.nvalid > 1 ? q.v.y : 0.f;
    r.z = q.nvalid > 2 ? q.v.z : 0.f;
    r.w = q.nvalid > 3 ? q.v.w : 0.f;
    return r;
}

// Workgroup shape: W waves, square tile TS = 32 W (64, 96 or 128): wave w owns rows [32w, 32w+32) of the tile and all
// of its TS columns (W accumulators).  The host picks W per launch: the per-(b,h) attention products have M = N = S
// (239..279), which a 128-tile pads to 256 or 384 but a 96-tile to 288; narrow outputs (the protein MLP, N = 242) get
// more, smaller workgroups so that every CU has work.
//
// Operand tile loader.  KC = true: the operand is K-contiguous (element (row,k) at base[row*srow + k]);
// thread t covers rows t/8 + 8W p (p < 4), k-quad (t%8)*4: 8 lanes read one full 128-byte line of a row.
// KC = false: row-contiguous (element at base[k*sk + row]); thread t covers k = t/(8W) + 8p, row-quad (t%(8W))*4.
template <bool KC, bool VEC, int W>
__device__ __forceinline__ void load_operand(const float* __restrict__ base, long long srow, long long sk, int row0,
                                             int nrows, int k0, int kend, int tid, Quad (&v)[4]) {
    if (KC) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = row0 + (tid >> 3) + 8 * W * p;
            const int nv = row < nrows ? kend - (k0 + kq) : 0;
            v[p] = load4<VEC>(base, (long long)row * srow + k0 + kq, nv);
        }
    } else {
        const int rq = (tid % (8 * W)) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int k = k0 + tid / (8 * W) + 8 * p;
            const int nv = k < kend ? nrows - (row0 + rq) : 0;
            v[p] = load4<VEC>(base, (long long)k * sk + row0 + rq, nv);
        }
    }
}

template <bool KC, int W>
__device__ __forceinline__ void store_operand(float* __restrict__ S, int tid, const Quad (&v)[4]) {
    if (KC) {
        const int kq = (tid & 7) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float4 m = masked(v[p]);
            float* d = S + ((tid >> 3) + 8 * W * p) * LDS_KC + kq;
            d[0] = m.x; d[1] = m.y; d[2] = m.z; d[3] = m.w;
        }
    } else {
        const int rq = (tid % (8 * W)) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            *reinterpret_cast<float4*>(S + (tid / (8 * W) + 8 * p) * (32 * W + 4) + rq) = masked(v[p]);
    }
}

template <bool KC, int W>
__device__ __forceinline__ float frag(const float* __restrict__ S, int row, int k) {
    return KC ? S[row * LDS_KC + k] : S[k * (32 * W + 4) + row];
}

#define GEMM_FOLD 4          // k-tiles (of BK) per accumulation chain
template <bool A_KC, bool B_KC, bool VA, bool VB, int W, bool FOLD = false>
__global__ void __launch_bounds__(64 * W, FOLD ? 2 : (W == 4 ? GEMM_WGS : 4))
gemm_f32_kernel(GemmArgs g) {
    constexpr int TS = 32 * W;
    __shared__ __attribute__((aligned(16))) float As[A_KC ? TS * LDS_KC : BK * (TS + 4)];
    __shared__ __attribute__((aligned(16))) float Bs[B_KC ? TS * LDS_KC : BK * (TS + 4)];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // Tile order: n fastest (the tiles that share an A row-panel run back to back), and the linear tile id is
    // remapped so that each XCD (workgroups b, b+8, ... share one) walks a contiguous chunk of that order: the
    // A panel is then fetched once into ONE L2 instead of eight.
    const int ntn = (g.N + TS - 1) / TS;
    const int nwg = gridDim.x;
    int pid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = pid & 7, loc = pid >> 3;
        pid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int m0 = (pid / ntn) * TS, n0 = (pid % ntn) * TS;
    const int z = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const int z1 = z / g.nb2, z2 = z % g.nb2;
    const float* A = g.A + z1 * g.sA1 + z2 * g.sA2;
    const float* B = g.B + z1 * g.sB1 + z2 * g.sB2;
    // K range of this split (multiples of BK)
    const int ktiles = (g.K + BK - 1) / BK;
    const int per = (ktiles + g.splitk - 1) / g.splitk;
    const int kbeg = ks * per * BK;
    const int kend = min(g.K, (ks + 1) * per * BK);
    const int nt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const int wm = wave * 32;
    const int i = lane & 31, h = lane >> 5;

    // Two-level summation: the MFMA accumulates its k-steps as one sequential fp32 chain; every GEMM_FOLD k-tiles the
    // chain is folded into `tot` and restarted.  A long chain of same-sign, near-identical addends (the attention
    // P.V product over ~750 identical padded tokens of a long-protein batch) otherwise picks up a rounding bias that
    // grows linearly with K (1.3e-5 at K = 1065, five times torch's blocked CPU sum); folded chains are <= 128 long.
    // FOLD instantiations (batched products with K > 512: the host picks them) pay 16 W registers for it.
    f32x16 acc[W], tot[FOLD ? W : 1];
#pragma unroll
    for (int b = 0; b < W; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[b][r] = 0.f;
            if (FOLD) tot[b][r] = 0.f;
        }

    // Register prefetch: tile t+1 is loaded into registers while tile t is multiplied out of LDS.
    Quad va[4], vb[4];
    if (nt > 0) {
        load_operand<A_KC, VA, W>(A, g.sam, g.sak, m0, g.M, kbeg, kend, tid, va);
        load_operand<B_KC, VB, W>(B, g.sbn, g.sbk, n0, g.N, kbeg, kend, tid, vb);
    }
    for (int t = 0; t < nt; ++t) {
        __syncthreads();
        store_operand<A_KC, W>(As, tid, va);
        store_operand<B_KC, W>(Bs, tid, vb);
        __syncthreads();
        if (t + 1 < nt) {
            load_operand<A_KC, VA, W>(A, g.sam, g.sak, m0, g.M, kbeg + (t + 1) * BK, kend, tid, va);
            load_operand<B_KC, VB, W>(B, g.sbn, g.sbk, n0, g.N, kbeg + (t + 1) * BK, kend, tid, vb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = frag<A_KC, W>(As, wm + i, kk + h);
            float bv[W];
#pragma unroll
            for (int b = 0; b < W; ++b) bv[b] = frag<B_KC, W>(Bs, 32 * b + i, kk + h);
#pragma unroll
            for (int b = 0; b < W; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv[b], acc[b], 0, 0, 0);
        }
        if (FOLD && (t % GEMM_FOLD) == GEMM_FOLD - 1) {
#pragma unroll
            for (int b = 0; b < W; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) { tot[b][r] += acc[b][r]; acc[b][r] = 0.f; }
        }
    }
    if (FOLD) {
#pragma unroll
        for (int b = 0; b < W; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] += tot[b][r];
    }

    float* C = g.C + z1 * g.sC1 + z2 * g.sC2;
    const float* R = g.R ? g.R + z1 * g.sC1 + z2 * g.sC2 : nullptr;
    const int mbase = m0 + wm;
#pragma unroll
    for (int b = 0; b < W; ++b) {
        const int n = n0 + 32 * b + i;
        const bool nok = n < g.N;
        const int nc = nok ? n : g.N - 1;
        const float bias = (g.bias && ks == 0) ? g.bias[nc] : 0.f;
        if (g.splitk > 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + mfma32_row(r, lane);
                if (m < g.M && nok) atomicAdd(&C[(long long)m * g.ldc + n], g.alpha * acc[b][r] + bias);
            }
        } else {
            float rv[16];
            if (R) {            // unconditional (clamped) residual loads: all 16 in flight together
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = min(mbase + mfma32_row(r, lane), g.M - 1);
                    rv[r] = R[(long long)m * g.ldr + nc];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + mfma32_row(r, lane);
                float v = g.alpha * acc[b][r] + bias;
                if (g.relu == 2) v = rv[r] > 0.f ? v : 0.f;      // ReLU backward: the "residual" is the mask source
                else v += rv[r];
                if (g.relu == 1) v = fmaxf(v, 0.f);
                if (m < g.M && nok) C[(long long)m * g.ldc + n] = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ aligned fast form
// For the big contractions of the joint-attention block (M, N multiples of 128, K of 32, 16-byte aligned operands, no
// batch): same 128x128x32 tiling and the same exact-fp32 MFMA, but
//   * operands go HBM/L2 -> LDS by global_load_lds_dwordx4 (no VGPR staging, no ds_write, no select VALU), into two
//     LDS stages that are two distinct arrays (one barrier per k-tile; the DMA of k-tile t+1 flies under the MFMAs of t);
//   * a K-contiguous operand lands as [row][32 k] with the 16-byte chunk c of row r at chunk position c ^ ((r>>1)&7)
//     (chosen on the SOURCE address; the LDS side of the DMA is lane-linear) and is read back with ONE conflict-free
//     ds_read_b128 per four MFMAs: the k order of the contraction is permuted (group g, step s uses k = 8g + 4h + s
//     for lane half h), identically for A and B, which only changes the fp32 summation order;
//   * a row-contiguous operand lands as [k][128 rows]; for A the wave's two 32-row tiles interleave rows (tile a owns
//     rows 2i + a) so both come from one ds_read_b64 (the permutation is undone in the epilogue for free);
//   * the grid is persistent: a workgroup walks its work items (output tile x K split) as ONE flat stream of k-tiles,
//     so the first k-tile of the next item is already in flight during the epilogue of the current one.
#ifndef FBK
#define FBK 16               // k-tile depth of the fast form
#endif
#ifndef FNST
#define FNST 3               // LDS stages (each its own array)
#endif
#ifndef FWGS
#define FWGS 2               // persistent workgroups per CU
#endif
#define FCH (FBK / 4)                    // 16-byte chunks per K-contiguous row
#define FRP (64 / FCH)                   // rows per 1 KiB DMA piece of a K-contiguous operand
#define FPCS (128 * FBK / 256 / 4)       // DMA pieces per wave per operand per k-tile
#define STAGE_FLOATS (2 * 128 * FBK)     // A tile then B tile

// 16 bytes per lane HBM/L2 -> LDS (lane l lands at l_base + 16 l).  Issued as inline asm on purpose: hipcc's
// waitcnt pass treats the builtin as an LDS write that may alias every later ds_read and, once the vmcnt waits are
// conditional (multi-stage pipeline), drains vmcnt(0) between the DMA issue and the fragment reads of the OTHER stage.
// Completion is tracked by hand (the s_waitcnt vmcnt in step()); the "memory" clobber keeps LDS reads on their side.
__device__ __forceinline__ void dma16(const float* g, unsigned lds) {     // lds: wave-uniform LDS byte address
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(g), "s"(lds) : "memory", "m0");
}
__device__ __forceinline__ unsigned lds_addr(const float* p) {            // LDS byte address of a __shared__ object
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)p);
}
__device__ __forceinline__ int fswz(int row) { return FBK == 32 ? (row >> 1) & 7 : (row >> 2) & 3; }

// Work items of the fast form.  Items [0, full) are whole output tiles (tile = item, all of K, plain epilogue).
// Items from `full` on are K-parts of the remaining `rem` tiles: part = (item - full) / rem covers k-tiles
// [part * per, ...) of tile full + (item - full) % rem and ADDS its partial sum to C with float atomics (part 0 also
// adds bias / residual).  Two uses: split-K weight gradients (full = 0, rem = all tiles, C accumulates), and the
// ragged last round of a persistent grid (the host zeroes those rows of C first) -- without it a [35k x 512] output
// (1116 tiles on 512 workgroups) costs 3 rounds for 2.2 rounds of work.
struct FastArgs {
    GemmArgs g;
    int ntm, ntn, items, nk_total, per, full, rem;
    int tpb;        // ragged / batched form: tiles per matrix of the batch (tile / tpb = matrix index, nb2 fastest)
};

template <bool KC>
struct OperandStream {                   // this lane's DMA source pointers into one operand + the k-tile step
    const float* p[FPCS];
    long long step;
    // last_row >= 0 (ragged form, K-contiguous operands only): rows past the extent re-read the last row; their
    // products are never stored
    template <bool CLAMP = false>
    __device__ __forceinline__ void set(const float* base, long long ld, int row0, int k0, int wave, int lane,
                                        int last_row = 0) {
        if (KC) {
#pragma unroll
            for (int pc = 0; pc < FPCS; ++pc) {
                const int row = (wave * FPCS + pc) * FRP + lane / FCH;
                const int grow = CLAMP ? min(row0 + row, last_row) : row0 + row;
                p[pc] = base + (long long)grow * ld + k0 + (((lane % FCH) ^ fswz(row)) << 2);
            }
            step = FBK;
        } else {
#pragma unroll
            for (int pc = 0; pc < FPCS; ++pc) {
                const int kr = (wave * FPCS + pc) * 2 + (lane >> 5);
                p[pc] = base + (long long)(k0 + kr) * ld + row0 + ((lane & 31) << 2);
            }
            step = FBK * ld;
        }
    }
    __device__ __forceinline__ void rewind(int nk) {
#pragma unroll
        for (int pc = 0; pc < FPCS; ++pc) p[pc] -= nk * step;
    }
    __device__ __forceinline__ void issue(unsigned tile, int wave) {      // tile: LDS byte address of the operand tile
#pragma unroll
        for (int pc = 0; pc < FPCS; ++pc) {
            dma16(p[pc], tile + (wave * FPCS + pc) * 1024);  // 1 KiB pieces: FRP rows x FBK k, or 2 k x 128 rows
            p[pc] += step;
        }
    }
};

#ifdef GEMM_CLOCK_PROBE
__device__ unsigned long long gemm_probe[4 * 1024];     // dev-only: per workgroup: start, end (100 MHz), cycles, hw id
#endif

// RAG (both operands K-contiguous): a batch of matrices (nb2, strides s?1 / s?2 of GemmArgs) whose M and N need not be
// multiples of 128 -- edge tiles read clamped rows and store guarded; used for the per-(b,h) attention products with
// K = 256 (scores = Q K^T and dP = dO V^T), which otherwise run on the generic form at 0.6 of this one's rate.
// X3: the k-tile's fp32 fragments (the same LDS reads) are split into three bf16 pieces each in registers and multiplied
// as six v_mfma_f32_32x32x16_bf16 per accumulator (see "fp32 from bf16 pieces" below): the operands still travel by LDS-DMA
// as fp32, nothing is stored to LDS from registers, and the whole persistent machinery (XCD order, split-K, ragged last
// round, column sums) is this kernel's.  The 8 k of a lane's bf16 fragment are the two 4-k groups the fp32 form reads for
// MFMA groups 0 and 1 -- the same permutation of k for A and B, which only changes the summation order.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
__device__ __forceinline__ void split8_regs(const float (&lo)[4], const float (&hi)[4], bf16x8_t (&p)[3]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? lo[j] : hi[j - 4];
        const __bf16 h = (__bf16)v;
        const float r1 = v - __uint_as_float(((unsigned)(*reinterpret_cast<const unsigned short*>(&h))) << 16);
        const __bf16 m = (__bf16)r1;
        const float r2 = r1 - __uint_as_float(((unsigned)(*reinterpret_cast<const unsigned short*>(&m))) << 16);
        p[0][j] = h;
        p[1][j] = m;
        p[2][j] = (__bf16)r2;
    }
}

template <bool A_KC, bool B_KC, bool RAG = false, bool X3 = false>
__global__ void __launch_bounds__(256, FWGS)
gemm_f32_fast_kernel(FastArgs fa) {
#ifdef GEMM_CLOCK_PROBE
    const unsigned long long pc0 = __builtin_amdgcn_s_memtime(), pr0 = __builtin_amdgcn_s_memrealtime();
#endif
    __shared__ __attribute__((aligned(16))) float stage0[STAGE_FLOATS];
    __shared__ __attribute__((aligned(16))) float stage1[STAGE_FLOATS];
#if FNST == 3
    __shared__ __attribute__((aligned(16))) float stage2[STAGE_FLOATS];
#endif
    const GemmArgs& g = fa.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int i = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int nwg = gridDim.x;
    int slot = blockIdx.x;               // workgroups b, b+8, ... share an XCD: give each XCD a contiguous run of items
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = slot & 7, loc = slot >> 3;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    if (slot >= fa.items) return;

    // work item -> tile origin and K range (in k-tiles)
    auto item_tile = [&](int it) { return it < fa.full ? it : fa.full + (it - fa.full) % fa.rem; };
    auto item_part = [&](int it) { return it < fa.full ? 0 : (it - fa.full) / fa.rem; };
    auto item_mat = [&](int it) { return RAG ? item_tile(it) / fa.tpb : 0; };
    auto item_loc = [&](int it) { return RAG ? item_tile(it) % fa.tpb : item_tile(it); };
    auto item_m0 = [&](int it) { return (item_loc(it) / fa.ntn) * 128; };
    auto item_n0 = [&](int it) { return (item_loc(it) % fa.ntn) * 128; };
    auto item_k0 = [&](int it) { return item_part(it) * fa.per; };
    auto item_nk = [&](int it) { return it < fa.full ? fa.nk_total : min(fa.per, fa.nk_total - item_part(it) * fa.per); };

    // loader state: the next k-tile to fetch
    OperandStream<A_KC> sa;
    OperandStream<B_KC> sb;
    // (Starting each workgroup at a different k-tile of its item -- so that neighbours read different columns of
    // their power-of-two-strided rows -- was measured and bought nothing; l_wrap stays 0.)
    auto item_rot = [&](int) { return 0; };
    int l_item = slot, l_left, l_wrap, l_nk;
    auto open_item = [&]() {
        l_nk = item_nk(l_item);
        l_left = l_nk;
        l_wrap = item_rot(l_item);
        const int k0 = (item_k0(l_item) + l_wrap) * FBK;
        if (RAG) {
            const int z = item_mat(l_item), z1 = z / g.nb2, z2 = z - z1 * g.nb2;
            sa.template set<true>(g.A + z1 * g.sA1 + z2 * g.sA2, g.sam, item_m0(l_item), k0, wave, lane, g.M - 1);
            sb.template set<true>(g.B + z1 * g.sB1 + z2 * g.sB2, g.sbn, item_n0(l_item), k0, wave, lane, g.N - 1);
        } else {
            sa.set(g.A, A_KC ? g.sam : g.sak, item_m0(l_item), k0, wave, lane);
            sb.set(g.B, B_KC ? g.sbn : g.sbk, item_n0(l_item), k0, wave, lane);
        }
    };
    open_item();
    const int uwave = __builtin_amdgcn_readfirstlane(wave);
    auto fetch = [&](const float* stage) -> bool {       // issue the DMA of the next k-tile (if any) into that stage
        if (l_item >= fa.items) return false;
        const unsigned st = lds_addr(stage);
        sa.issue(st, uwave);
        sb.issue(st + 128 * FBK * 4, uwave);
        if (--l_left == 0) {
            l_item += nwg;
            if (l_item < fa.items) open_item();
        } else if (l_left == l_wrap) {
            sa.rewind(l_nk);
            sb.rewind(l_nk);
        }
        return true;
    };

    // column sums of A (= the bias gradient when A is dY^T): the waves that hold column tile 0 add their A fragments up
    float cs0 = 0.f, cs1 = 0.f;
    bool do_cs = !A_KC && g.csum != nullptr && wn == 0 && item_n0(slot) == 0;
    f32x16 acc[2][2];
    auto zero_acc = [&]() {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    };
    zero_acc();

    // fragment read offsets (floats) inside a stage
    const int sw = fswz(i);
    const int a_off = A_KC ? (wm + i) * FBK : wm + 2 * i;
    const int b_off = 128 * FBK + (B_KC ? (wn + i) * FBK : wn + i);

    struct Frag { float a0[4], a1[4], b0[4], b1[4]; };
    auto load_frag = [&](Frag& f, const float* st, int gq) {
        if (A_KC) {
            const int c = (((2 * gq + h) ^ sw) << 2);
            const float4 x0 = *reinterpret_cast<const float4*>(st + a_off + c);
            const float4 x1 = *reinterpret_cast<const float4*>(st + a_off + 32 * FBK + c);
            f.a0[0] = x0.x; f.a0[1] = x0.y; f.a0[2] = x0.z; f.a0[3] = x0.w;
            f.a1[0] = x1.x; f.a1[1] = x1.y; f.a1[2] = x1.z; f.a1[3] = x1.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float2 x = *reinterpret_cast<const float2*>(st + a_off + (8 * gq + 4 * h + s) * 128);
                f.a0[s] = x.x; f.a1[s] = x.y;
            }
        }
        if (B_KC) {
            const int c = (((2 * gq + h) ^ sw) << 2);
            const float4 y0 = *reinterpret_cast<const float4*>(st + b_off + c);
            const float4 y1 = *reinterpret_cast<const float4*>(st + b_off + 32 * FBK + c);
            f.b0[0] = y0.x; f.b0[1] = y0.y; f.b0[2] = y0.z; f.b0[3] = y0.w;
            f.b1[0] = y1.x; f.b1[1] = y1.y; f.b1[2] = y1.z; f.b1[3] = y1.w;
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                f.b0[s] = st[b_off + (8 * gq + 4 * h + s) * 128];
                f.b1[s] = st[b_off + (8 * gq + 4 * h + s) * 128 + 32];
            }
        }
    };
    auto mfma16 = [&](const Frag& f) {
        if (!A_KC && do_cs) {
            cs0 += (f.a0[0] + f.a0[1]) + (f.a0[2] + f.a0[3]);
            cs1 += (f.a1[0] + f.a1[1]) + (f.a1[2] + f.a1[3]);
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b0[s], acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[s], f.b1[s], acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b0[s], acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[s], f.b1[s], acc[1][1], 0, 0, 0);
        }
    };
    // fragments of group g+1 are read while group g is in the MFMA
    // (sched_barrier pins the order: hipcc otherwise sinks each read group down to its use)
    auto compute = [&](const float* st) {
        Frag f0, f1;
        load_frag(f0, st, 0);
        load_frag(f1, st, 1);
        if (X3) {
            // the bf16-pieces body consumes exactly the two 8-k fragment groups of a 16-deep k-tile
            static_assert(!X3 || FBK == 16, "gemm_f32_fast_kernel<X3>: the bf16-pieces branch needs FBK == 16");
            if (!A_KC && do_cs) {
                cs0 += ((f0.a0[0] + f0.a0[1]) + (f0.a0[2] + f0.a0[3])) + ((f1.a0[0] + f1.a0[1]) + (f1.a0[2] + f1.a0[3]));
                cs1 += ((f0.a1[0] + f0.a1[1]) + (f0.a1[2] + f0.a1[3])) + ((f1.a1[0] + f1.a1[1]) + (f1.a1[2] + f1.a1[3]));
            }
            bf16x8_t pa[2][3], pb[2][3];
            split8_regs(f0.a0, f1.a0, pa[0]);
            split8_regs(f0.a1, f1.a1, pa[1]);
            split8_regs(f0.b0, f1.b0, pb[0]);
            split8_regs(f0.b1, f1.b1, pb[1]);
            // small terms first; the four accumulators take turns, so consecutive MFMAs are independent
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int t = 0; t < 6; ++t)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[a][PA[t]], pb[b][PB[t]], acc[a][b], 0, 0, 0);
            return;
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma16(f0);
        __builtin_amdgcn_sched_barrier(0);
        if (FBK == 32) {
            load_frag(f0, st, 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        mfma16(f1);
        __builtin_amdgcn_sched_barrier(0);
        if (FBK == 32) {
            load_frag(f1, st, 3);
            __builtin_amdgcn_sched_barrier(0);
            mfma16(f0);
            mfma16(f1);
        }
    };

    auto epilogue = [&](int it) {
        const int m0 = item_m0(it), n0 = item_n0(it);
        const bool first_split = item_part(it) == 0;
        const bool partial = it >= fa.full;
        float* Cm = g.C;
        if (RAG) {
            const int z = item_mat(it), z1 = z / g.nb2, z2 = z - z1 * g.nb2;
            Cm += z1 * g.sC1 + z2 * g.sC2;
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int n = n0 + wn + 32 * b + i;
                if (RAG) {                                   // guarded plain stores (alpha, optional bias; nothing else)
                    const float bias = (g.bias && n < g.N) ? g.bias[n] : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + wm + 32 * a + mfma32_row(r, lane);
                        if (m < g.M && n < g.N) Cm[(long long)m * g.ldc + n] = g.alpha * acc[a][b][r] + bias;
                    }
                    continue;
                }
                const float bias = (g.bias && first_split) ? g.bias[n] : 0.f;
                float* crow = g.C + n;
                if (partial) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mfma32_row(r, lane);
                        const int m = m0 + wm + (A_KC ? 32 * a + rr : 2 * rr + a);
                        float v = g.alpha * acc[a][b][r] + bias;
                        if (g.R && first_split) v += g.R[(long long)m * g.ldr + n];
                        atomicAdd(crow + (long long)m * g.ldc, v);
                    }
                } else {
                    float rv[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mfma32_row(r, lane);
                        const int m = m0 + wm + (A_KC ? 32 * a + rr : 2 * rr + a);
                        rv[r] = g.R ? g.R[(long long)m * g.ldr + n] : 0.f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rr = mfma32_row(r, lane);
                        const int m = m0 + wm + (A_KC ? 32 * a + rr : 2 * rr + a);
                        float v = g.alpha * acc[a][b][r] + bias;
                        if (g.relu == 2) v = rv[r] > 0.f ? v : 0.f;
                        else v += rv[r];
                        if (g.relu == 1) v = fmaxf(v, 0.f);
                        crow[(long long)m * g.ldc] = v;
                    }
                }
            }
    };

    int c_item = slot, c_left = item_nk(slot);
    bool ahead = false;                  // a younger DMA (the stage after this one) is in flight and nothing else is
    fetch(stage0);
#if FNST == 3
    ahead = fetch(stage1);
#endif
    // one k-tile: wait for its DMA, barrier (every wave is past its reads of the stage about to be refilled), start
    // the DMA of the k-tile FNST-1 ahead, multiply; at the end of an item write it out.  false = stream done.
    auto step = [&](float* cur, float* refill) -> bool {
        // vmcnt: this stage's DMA has landed.  With 3 stages the next stage's 2*FPCS requests may stay in flight
        // (requests retire in order); after an epilogue everything is drained (its stores are younger than them).
        if (FNST == 3 && ahead) __builtin_amdgcn_s_waitcnt(0x0F70 | (2 * FPCS));
        else __builtin_amdgcn_s_waitcnt(0x0F70);
        __builtin_amdgcn_s_barrier();    // (bare barrier: __syncthreads() would add a vmcnt(0) fence)
        asm volatile("" ::: "memory");
        ahead = fetch(refill);
        compute(cur);
        if (--c_left == 0) {
            epilogue(c_item);
            if (!A_KC && do_cs) {                    // lanes i and i + 32 hold the two k halves of rows wm + 2 i (+ 1)
                const float t0 = cs0 + __shfl_xor(cs0, 32, 64), t1 = cs1 + __shfl_xor(cs1, 32, 64);
                if (h == 0) {
                    atomicAdd(&g.csum[item_m0(c_item) + wm + 2 * i], t0);
                    atomicAdd(&g.csum[item_m0(c_item) + wm + 2 * i + 1], t1);
                }
                cs0 = cs1 = 0.f;
            }
            ahead = false;
            c_item += nwg;
            if (c_item >= fa.items) return false;
            c_left = item_nk(c_item);
            do_cs = !A_KC && g.csum != nullptr && wn == 0 && item_n0(c_item) == 0;
            zero_acc();
        }
        return true;
    };
    while (true) {
#if FNST == 3
        if (!step(stage0, stage2)) break;
        if (!step(stage1, stage0)) break;
        if (!step(stage2, stage1)) break;
#else
        if (!step(stage0, stage1)) break;
        if (!step(stage1, stage0)) break;
#endif
    }
#ifdef GEMM_CLOCK_PROBE
    if (threadIdx.x == 0 && blockIdx.x < 1024) {
        gemm_probe[4 * blockIdx.x + 0] = pr0;
        gemm_probe[4 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        gemm_probe[4 * blockIdx.x + 2] = __builtin_amdgcn_s_memtime() - pc0;
        gemm_probe[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)) |    // HW_ID
                                         ((unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32);  // XCC_ID
    }
#endif
}

// ------------------------------------------------------------------------------------------------ fp32 from bf16 pieces
// C = alpha A B (+ bias, residual, ReLU / ReLU mask; split-K parts add with float atomics), fp32 in, fp32 out, at fp32
// accuracy on the bf16 matrix pipe (16 x the fp32 MFMA rate).  Every operand element is split on the fly into three
// bf16 pieces x = h + m + l (round-to-nearest at each level; the residual of a level is exact in fp32, so |x - (h+m+l)|
// <= 2^-24 |x|) and a product keeps the six piece products of weight >= 2^-16: hh, hm, mh, mm, hl, lh -- each exact in
// the fp32 accumulator of v_mfma_f32_32x32x16_bf16 -- and drops ml, lm, ll (<= 2^-23 of the product, the size of the
// rounding of one fp32 FMA).  Measured against fp64 on the FFN shapes it is as close as an fp32 FMA chain
// (tools/scratch/gemm_bf16x3_lab.hip: K = 512: 5.9e-7 against 8.2e-7 of max|C|; K = 2048: 2.0e-6 against 1.5e-6).
// Used for the BACKWARD products only (dX = dY W and dW = dY^T X: B row-contiguous): the forward products stay on the fp32
// MFMA kernels, so every forward value -- and with it every ReLU mask -- is what it was.
// 128 x 128 output tile, 4 waves of 64 x 64, k16 slabs through two LDS buffers: global (fp32) -> registers (two slabs in
// flight) -> split -> LDS ([buffer][piece][row][k]) in the same basic block as the MFMAs of the slab before, one barrier per
// slab; a row-contiguous operand is read as 4 (k) x 2 (rows) blocks per thread and transposed in registers on its way.
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
constexpr int XBK = 32;

struct X3Args {
    GemmArgs g;
    int ntm, ntn, nk_total, per, parts;          // k-tiles in all, per split part; parts > 1: C accumulates (atomics)
};

__device__ __forceinline__ float bf16_up(__bf16 v) {
    return __uint_as_float(((unsigned)(*reinterpret_cast<unsigned short*>(&v))) << 16);
}
__device__ __forceinline__ void split4(const float v[4], bf16x4& h, bf16x4& m, bf16x4& l) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const __bf16 hi = (__bf16)v[i];
        const float r1 = v[i] - bf16_up(hi);
        const __bf16 mi = (__bf16)r1;
        h[i] = hi;
        m[i] = mi;
        l[i] = (__bf16)(r1 - bf16_up(mi));
    }
}

// LDS: [buffer][piece][row][k16 slab], 48-byte rows; the two 16-byte halves of a row trade places by bits 2 ^ 3 ^ 4 of
// the row: the fragment reads (ds_read_b128's non-contiguous lane groups) are conflict-free (round 3; until then the
// halves traded places on rows 16..31 and the reads were two-way, like the transposing stores of a row-contiguous
// operand, which still are in this kernel: a wave stores one k block, so only every other 8-byte slot of the 128-byte
// window is ever used.  The thread map that cures that -- 4 row pairs x 4 k blocks per 16 lanes, gemm_bf16x3_dw_kernel --
// measured slower HERE: 5.15 against 4.87 ms per step with every weight gradient on this kernel; its global loads then
// touch four k rows per instruction.)
constexpr int XLD16 = 24;
__device__ __forceinline__ int x3_half(int row, int half) { return (half ^ (((row >> 2) ^ (row >> 3) ^ (row >> 4)) & 1)) * 8; }

template <bool KC>
struct X3Operand {                      // one 128 x 16 operand slab: global -> registers -> three bf16 images in LDS
    // KC (element (row, k) at row * ld + k): thread = (row, half): 8 consecutive k of one row.
    // !KC (element (row, k) at k * ld + row): thread = (row pair, k block): 4 k x 2 rows, transposed in registers.
    __device__ static __forceinline__ void load(float (&r)[8], const float* base, long long ld, int row0, int k0, int tid) {
        if (KC) {
            const float* p = base + (long long)(row0 + (tid >> 1)) * ld + k0 + 8 * (tid & 1);
            *reinterpret_cast<float4*>(&r[0]) = *reinterpret_cast<const float4*>(p);
            *reinterpret_cast<float4*>(&r[4]) = *reinterpret_cast<const float4*>(p + 4);
        } else {
            const float* p = base + (long long)(k0 + 4 * (tid >> 6)) * ld + row0 + 2 * (tid & 63);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<float2*>(&r[2 * kk]) = *reinterpret_cast<const float2*>(p + kk * ld);
        }
    }
    __device__ static __forceinline__ void store(const float (&r)[8], __bf16 (*s)[128][XLD16], int tid) {
        if (KC) {
            const int row = tid >> 1, c = x3_half(row, tid & 1);
            bf16x4 a0, a1, a2, b0, b1, b2;
            split4(&r[0], a0, a1, a2);
            split4(&r[4], b0, b1, b2);
            *reinterpret_cast<bf16x8*>(&s[0][row][c]) = __builtin_shufflevector(a0, b0, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(&s[1][row][c]) = __builtin_shufflevector(a1, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            *reinterpret_cast<bf16x8*>(&s[2][row][c]) = __builtin_shufflevector(a2, b2, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            const int kb = tid >> 6;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = 2 * (tid & 63) + j;
                const float v[4] = {r[j], r[2 + j], r[4 + j], r[6 + j]};
                bf16x4 p0, p1, p2;
                split4(v, p0, p1, p2);
                const int c = x3_half(row, kb >> 1) + 4 * (kb & 1);
                *reinterpret_cast<bf16x4*>(&s[0][row][c]) = p0;
                *reinterpret_cast<bf16x4*>(&s[1][row][c]) = p1;
                *reinterpret_cast<bf16x4*>(&s[2][row][c]) = p2;
            }
        }
    }
};

template <bool A_KC, bool B_KC>
__global__ void __launch_bounds__(256, 2)
gemm_bf16x3_kernel(X3Args xa) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][128][XLD16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][128][XLD16];
    __shared__ float s_cs[128];
    const GemmArgs& g = xa.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int tiles = xa.ntm * xa.ntn;
    int slot = blockIdx.x;               // workgroups b, b + 8, ... share an XCD: give each XCD a contiguous run of work items
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = slot & 7, loc = slot >> 3;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile = slot % tiles, part = slot / tiles;
    const int m0 = (tile / xa.ntn) * 128, n0 = (tile % xa.ntn) * 128;
    const int kt0 = part * xa.per, nk = min(xa.per, xa.nk_total - kt0);
    const int kbase = kt0 * XBK, ns = 2 * nk;               // k16 slabs of this part (even)
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const bool do_cs = !A_KC && g.csum != nullptr && n0 == 0;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float cs[2] = {0.f, 0.f};
    if (tid < 128) s_cs[tid] = 0.f;
    const long long lda = A_KC ? g.sam : g.sak, ldb = B_KC ? g.sbn : g.sbk;
    float ra[2][8], rb[2][8];                               // two slabs in flight (slot = slab & 1)
    // One slab: the split + store of slab s+1 and the request for slab s+3 sit in the SAME basic block as the MFMAs of
    // slab s (no branch in the body: past the end the stale registers go into the idle buffer and the last slab is asked
    // for again), one barrier per slab.
#define X3_LOAD(SL, s_)                                                                       \
    do {                                                                                      \
        X3Operand<A_KC>::load(ra[SL], g.A, lda, m0, kbase + 16 * (s_), tid);                  \
        X3Operand<B_KC>::load(rb[SL], g.B, ldb, n0, kbase + 16 * (s_), tid);                  \
    } while (0)
#define X3_STORE(SL, BUF, s_)                                                                 \
    do {                                                                                      \
        if (!A_KC) {                                                                          \
            const bool live = do_cs && (s_) < ns;                                             \
            cs[0] += live ? (ra[SL][0] + ra[SL][2]) + (ra[SL][4] + ra[SL][6]) : 0.f;          \
            cs[1] += live ? (ra[SL][1] + ra[SL][3]) + (ra[SL][5] + ra[SL][7]) : 0.f;          \
        }                                                                                     \
        X3Operand<A_KC>::store(ra[SL], sA[BUF], tid);                                         \
        X3Operand<B_KC>::store(rb[SL], sB[BUF], tid);                                         \
    } while (0)
#define X3_STEP(CUR, s_)                                                                      \
    do {                                                                                      \
        X3_STORE(1 - CUR, 1 - CUR, (s_) + 1);                                                 \
        X3_LOAD(1 - CUR, min((s_) + 3, ns - 1));                                              \
        bf16x8 fa[3][2], fb[3][2];                                                            \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                         \
            _Pragma("unroll") for (int t = 0; t < 2; ++t) {                                   \
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[CUR][p][wm + 32 * t + i][x3_half(wm + 32 * t + i, h)]); \
                fb[p][t] = *reinterpret_cast<const bf16x8*>(&sB[CUR][p][wn + 32 * t + i][x3_half(wn + 32 * t + i, h)]); \
            }                                                                                 \
        /* small terms first; the four accumulators take turns, so consecutive MFMAs are independent */ \
        _Pragma("unroll") for (int t = 0; t < 6; ++t)                                         \
            _Pragma("unroll") for (int a = 0; a < 2; ++a)                                     \
                _Pragma("unroll") for (int b = 0; b < 2; ++b)                                 \
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[t]][a], fb[PB[t]][b], acc[a][b], 0, 0, 0); \
        __syncthreads();                                                                      \
    } while (0)
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    X3_LOAD(0, 0);
    X3_LOAD(1, 1);
    X3_STORE(0, 0, 0);
    X3_LOAD(0, min(2, ns - 1));
    __syncthreads();
    for (int s_i = 0; s_i < ns; s_i += 2) {
        X3_STEP(0, s_i);
        X3_STEP(1, s_i + 1);
    }
#undef X3_LOAD
#undef X3_STORE
#undef X3_STEP
    const bool first = part == 0, partial = g.splitk > 1;       // split-K by contract ADDS onto C, even as a single part
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int n = n0 + wn + 32 * b + i;
            const float bias = (g.bias && first) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + mfma32_row(r, lane);
                float v = g.alpha * acc[a][b][r] + bias;
                if (partial) {
                    if (g.R && first) v += g.R[(long long)m * g.ldr + n];
                    atomicAdd(&g.C[(long long)m * g.ldc + n], v);
                } else {
                    const float rv = g.R ? g.R[(long long)m * g.ldr + n] : 0.f;
                    if (g.relu == 2) v = rv > 0.f ? v : 0.f;
                    else v += rv;
                    if (g.relu == 1) v = fmaxf(v, 0.f);
                    g.C[(long long)m * g.ldc + n] = v;
                }
            }
        }
    if (do_cs) {                                 // (A row-contiguous: thread (row pair, k block) carried rows 2 p, 2 p + 1)
        atomicAdd(&s_cs[2 * (tid & 63)], cs[0]);
        atomicAdd(&s_cs[2 * (tid & 63) + 1], cs[1]);
        __syncthreads();
        if (tid < 128) atomicAdd(&g.csum[m0 + tid], s_cs[tid]);
    }
}

// Weight gradients (both operands row-contiguous: dW = dY^T X) on 256 x 256 workgroup tiles, eight waves of 64 x 128.
// The register-staged kernel above splits every element of dY once per 128-column tile of dW and every element of X once
// per 128-row tile: per k16 slab a thread splits 16 elements for its wave's 24 MFMAs, as many VALU cycles as matrix
// cycles (the same balance the PMC passes showed for the in-kernel forward form).  With 256 x 256 tiles a thread still
// splits 16 elements per slab but its wave has 48 MFMAs for them: half the split work per product.  Same LDS images,
// same staging (thread = row pair x 4 k, transposed in registers), same piece order as gemm_bf16x3_kernel<false, false>;
// one workgroup per CU (147 KiB of LDS), the contraction split over 256 / tiles workgroups, partial sums added with
// float atomics (the dW contract: C accumulates); the bias gradient is summed from the fp32 registers before the split.
// LDS image: [buffer][piece][row][16 k], 48-byte rows; the 8-byte chunk q (k = 4 q .. 4 q + 3) of row r sits at chunk
// q ^ dw_swz(r) (dw_swz is 0 or 2: the two 16-byte halves of a row trade places; the order of k inside a fragment must be
// the same in every lane, an MFMA pairs every A row with every B column).  PMC on the first version (the images and the
// thread map of gemm_bf16x3_kernel) read SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE: the transposing 8-byte stores
// of 16 consecutive row pairs (one k block per wave: only every other 8-byte slot of the 128-byte window is ever used)
// and the 16-byte fragment reads of ds_read_b128's non-contiguous lane groups were both two-way.  Here a 16-lane group
// stages 4 row pairs x 4 k blocks and the halves trade places by bits 2 ^ 3 ^ 4 of the row: both conflict-free (checked
// with the bank rules of MI355X_MICROARCH.md, then with the counter: 0).
__device__ __forceinline__ int dw_swz(int row) { return (((row >> 2) ^ (row >> 3) ^ (row >> 4)) & 1) << 1; }

#ifdef DW_CLOCK
// lab build (-DDW_CLOCK, tools/scratch/dw_time.py): per workgroup, shader cycles (s_memtime) and 100 MHz ticks
// (s_memrealtime) from entry to the end of the slab loop, and the ticks at entry / loop begin / loop end / exit.  What it
// showed on the FFN gradient (S = 221): 4 409 cycles per slab against 3 072 of matrix-pipe time, at 1.66 GHz on normally
// distributed operands and 2.29 GHz on all-zero ones (same instruction stream: the launch is power-limited).
__device__ unsigned long long dw_clk[1024 * 4];
__device__ unsigned long long dw_clk2[1024 * 4];
extern "C" int jova_debug_dw_clk(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dw_clk), sizeof(unsigned long long) * n);
}
extern "C" int jova_debug_dw_clk2(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(dw_clk2), sizeof(unsigned long long) * n);
}
#endif
__global__ void __launch_bounds__(512, 1)
gemm_bf16x3_dw_kernel(X3Args xa) {
    __shared__ __attribute__((aligned(16))) __bf16 sA[2][3][256][XLD16];
    __shared__ __attribute__((aligned(16))) __bf16 sB[2][3][256][XLD16];
    __shared__ float s_cs[256];
    const GemmArgs& g = xa.g;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int tiles = xa.ntm * xa.ntn;
    int slot = blockIdx.x;               // workgroups b, b + 8, ... share an XCD: give each XCD a contiguous run of work items
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = slot & 7, loc = slot >> 3;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    const int tile = slot % tiles, part = slot / tiles;
    const int m0 = (tile / xa.ntn) * 256, n0 = (tile % xa.ntn) * 256;
    const int kt0 = part * xa.per, nk = min(xa.per, xa.nk_total - kt0);
    const int kbase = kt0 * XBK, ns = 2 * nk;               // k16 slabs of this part (even)
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 128;
    const bool do_cs = g.csum != nullptr && n0 == 0;
#ifdef DW_CLOCK
    const unsigned long long clk_t0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    f32x16 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float cs[2] = {0.f, 0.f};
    if (tid < 256) s_cs[tid] = 0.f;
    const long long lda = g.sak, ldb = g.sbk;
    const int rp = tid >> 2, kq = tid & 3;                  // this thread stages rows 2 rp, 2 rp + 1 at k = 4 kq .. 4 kq + 3
    float ra[2][8], rb[2][8];                               // two slabs in flight (slot = slab & 1)
    auto load_slab = [&](float (&r)[8], const float* base, long long ld, int row0, int k0) {
        const float* p = base + (long long)(k0 + 4 * kq) * ld + row0 + 2 * rp;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) *reinterpret_cast<float2*>(&r[2 * kk]) = *reinterpret_cast<const float2*>(p + kk * ld);
    };
    auto store_slab = [&](const float (&r)[8], __bf16 (*sp)[256][XLD16]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = 2 * rp + j;
            const float v[4] = {r[j], r[2 + j], r[4 + j], r[6 + j]};
            bf16x4 p0, p1, p2;
            split4(v, p0, p1, p2);
            const int c = 4 * (kq ^ dw_swz(row));
            *reinterpret_cast<bf16x4*>(&sp[0][row][c]) = p0;
            *reinterpret_cast<bf16x4*>(&sp[1][row][c]) = p1;
            *reinterpret_cast<bf16x4*>(&sp[2][row][c]) = p2;
        }
    };
#define DW_LOAD(SL, s_)                                                                       \
    do {                                                                                      \
        load_slab(ra[SL], g.A, lda, m0, kbase + 16 * (s_));                                   \
        load_slab(rb[SL], g.B, ldb, n0, kbase + 16 * (s_));                                   \
    } while (0)
#define DW_STORE(SL, BUF, s_)                                                                 \
    do {                                                                                      \
        const bool live = do_cs && (s_) < ns;                                                 \
        cs[0] += live ? (ra[SL][0] + ra[SL][2]) + (ra[SL][4] + ra[SL][6]) : 0.f;              \
        cs[1] += live ? (ra[SL][1] + ra[SL][3]) + (ra[SL][5] + ra[SL][7]) : 0.f;              \
        store_slab(ra[SL], sA[BUF]);                                                          \
        store_slab(rb[SL], sB[BUF]);                                                          \
    } while (0)
    // One slab: the split + store of slab s+1 and the request for slab s+3 sit in the same basic block as the MFMAs of
    // slab s, one barrier per slab.  Fragments: all three pieces of the wave's two A row blocks, the four B column blocks
    // one piece at a time (small terms first: A.h B.l | A.m B.m, A.h B.m | A.l B.h, A.m B.h, A.h B.h).
#define DW_STEP(CUR, s_)                                                                      \
    do {                                                                                      \
        DW_STORE(1 - CUR, 1 - CUR, (s_) + 1);                                                 \
        DW_LOAD(1 - CUR, min((s_) + 3, ns - 1));                                              \
        bf16x8 fa[3][2];                                                                      \
        _Pragma("unroll") for (int p = 0; p < 3; ++p)                                         \
            _Pragma("unroll") for (int t = 0; t < 2; ++t)                                     \
                fa[p][t] = *reinterpret_cast<const bf16x8*>(&sA[CUR][p][wm + 32 * t + i][8 * (h ^ (dw_swz(i) >> 1))]); \
        _Pragma("unroll") for (int pb = 2; pb >= 0; --pb) {                                   \
            bf16x8 fb[4];                                                                     \
            _Pragma("unroll") for (int t = 0; t < 4; ++t)                                     \
                fb[t] = *reinterpret_cast<const bf16x8*>(&sB[CUR][pb][wn + 32 * t + i][8 * (h ^ (dw_swz(i) >> 1))]); \
            _Pragma("unroll") for (int pa = 2 - pb; pa >= 0; --pa)                            \
                _Pragma("unroll") for (int a = 0; a < 2; ++a)                                 \
                    _Pragma("unroll") for (int b = 0; b < 4; ++b)                             \
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][a], fb[b], acc[a][b], 0, 0, 0); \
        }                                                                                     \
        __syncthreads();                                                                      \
    } while (0)
    DW_LOAD(0, 0);
    DW_LOAD(1, 1);
    DW_STORE(0, 0, 0);
    DW_LOAD(0, min(2, ns - 1));
    __syncthreads();
#ifdef DW_CLOCK
    const unsigned long long clk_r1 = __builtin_amdgcn_s_memrealtime();
#endif
    for (int s_i = 0; s_i < ns; s_i += 2) {
        DW_STEP(0, s_i);
        DW_STEP(1, s_i + 1);
    }
#undef DW_LOAD
#undef DW_STORE
#undef DW_STEP
#ifdef DW_CLOCK
    const unsigned long long clk_r2 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned long long* d = dw_clk + 4 * blockIdx.x;
        d[0] = __builtin_amdgcn_s_memtime() - clk_t0;
        d[1] = clk_r2 - clk_r0;
        d[2] = (unsigned long long)ns;
        d[3] = clk_r0;
    }
#endif
    const bool first = part == 0;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int n = n0 + wn + 32 * b + i;
            const float bias = (g.bias && first) ? g.bias[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm + 32 * a + mfma32_row(r, lane);
                float v = g.alpha * acc[a][b][r] + bias;
                if (g.R && first) v += g.R[(long long)m * g.ldr + n];
                atomicAdd(&g.C[(long long)m * g.ldc + n], v);
            }
        }
    if (do_cs) {                                 // thread (row pair, k block) carried rows 2 rp, 2 rp + 1
        atomicAdd(&s_cs[2 * rp], cs[0]);
        atomicAdd(&s_cs[2 * rp + 1], cs[1]);
        __syncthreads();
        if (tid < 256) atomicAdd(&g.csum[m0 + tid], s_cs[tid]);
    }
#ifdef DW_CLOCK
    __syncthreads();
    if (tid == 0 && blockIdx.x < 1024) {
        unsigned long long* d = dw_clk2 + 4 * blockIdx.x;
        d[0] = clk_r0; d[1] = clk_r1; d[2] = clk_r2; d[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ------------------------------------------------------------------------------------------------ bf16 pieces, B split beforehand
// The bf16-pieces product for a K-contiguous A and a B that many row tiles share (B = a weight matrix: the forward
// products x W^T and the data gradients dY W of the FFN / projection layers).  PMC passes over the in-kernel form
// (profiles/r03_pmc_x3_inkernel_ffn1.json) show its limit: 7.9 VALU instructions per MFMA -- the split of 32 floats per
// lane and k-tile, every element of A split by both waves that share its rows and every element of B by both waves that
// share its columns AND by every one of the M / 128 row tiles -- as many VALU cycles as matrix cycles (434 M against
// 439 M), of which only a quarter overlap.  Here
//   * B is split ONCE per call by gemm_b_image_kernel into an image of bf16 MFMA fragments (6 bytes per element, in the
//     order the waves read it: [column tile][k32 block][piece][16-column block][lane][8 bf16]; lane l = column l & 15,
//     k = 8 (l >> 4) + j), which the workgroups stream into LDS by global_load_lds_dwordx4, 24 KiB per k32 block, three
//     stages (B is requested two blocks ahead), one barrier per block;
//   * a wave owns 64 rows x 128 columns of the workgroup's 256 x 128 tile, so its A rows are its own: they come straight
//     from global memory into registers (32 contiguous bytes per lane, row block and k32 block, requested one block
//     ahead; no LDS, no barrier) and each element is split once -- 1.8 VALU instructions per MFMA-equivalent;
//   * v_mfma_f32_16x16x32_bf16 (four 16-row blocks x eight 16-column blocks of accumulators per wave): one MFMA
//     contracts the whole k32 block, so A travels and is split in ROW halves (blocks 0, 1 then 2, 3; its own registers,
//     wait and split per half, which keeps the pieces of only one half alive beside the 128 accumulator registers).
//     The chip holds a higher clock on this shape than on 32x32x16 under a matrix-dense load (MI355X_MICROARCH.md, DVFS
//     item 7): measured here on the Davis step, 302 -> 278 us per FFN product (profiles/r03_x3w_forms.json, which also
//     holds the phase stamps of the two earlier forms that led here: 128 x 128 tiles were bound by their ten
//     vector-memory instructions per 48 MFMAs and their four-byte stores);
//   * the image fragment is the MFMA's A operand and the activation piece its B operand (C^T = W x^T), so a lane ends
//     up with four consecutive columns of one row; the epilogue turns each wave's block round in LDS and stores whole
//     128-byte row segments.
// Vector memory is inline asm with hand-kept s_waitcnt vmcnt (requests retire in order).  Order per block s: [top]
// B(s+2) (6 requests), H0(s+1) (4) ... [middle] H1(s+1) (4); the top needs H0(s) and B(s): younger than H0(s) is only
// H1(s) -> vmcnt(4); the middle needs H1(s): younger are B(s+2) and H0(s+1) -> vmcnt(10 | 4 | 0).
// Rows past M (M % 256 == 128: the last row tile is half empty) read row M - 1 and are not stored.
struct X3WArgs {
    GemmArgs g;
    const unsigned char* image;
    int ntm, ntn, nkb;
    int vec_c;                  // C, the residual and the bias take 16-byte accesses
};
constexpr int X3W_STAGE = 24 * 1024;
#ifndef X3W_NG
#define X3W_NG 4             // column tiles per group: an XCD's resident workgroups share X3W_NG tiles' images (L2)
#endif

__global__ void __launch_bounds__(256)
gemm_b_image_kernel(const float* __restrict__ B, long long sbk, long long sbn, int N, int K, uint4* __restrict__ image) {
    // one thread per (column, 8 k): unit u = ((nt * nkb + kb) * 8 + nb) * 64 + lane; the three pieces lie 8 KiB apart
    const long long u = (long long)blockIdx.x * 256 + threadIdx.x;
    if (u >= (long long)N * K / 8) return;
    const int lane = (int)(u & 63), nb = (int)((u >> 6) & 7);
    const long long blk = u >> 9;
    const int nkb = K / 32;
    const int kb = (int)(blk % nkb), nt = (int)(blk / nkb);
    const int n = nt * 128 + nb * 16 + (lane & 15), k = kb * 32 + 8 * (lane >> 4);
    float lo[4], hi[4];
    if (sbk == 1) {
        const float4 x = *reinterpret_cast<const float4*>(B + (long long)n * sbn + k);
        const float4 y = *reinterpret_cast<const float4*>(B + (long long)n * sbn + k + 4);
        lo[0] = x.x; lo[1] = x.y; lo[2] = x.z; lo[3] = x.w;
        hi[0] = y.x; hi[1] = y.y; hi[2] = y.z; hi[3] = y.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            lo[j] = B[(long long)(k + j) * sbk + n];
            hi[j] = B[(long long)(k + 4 + j) * sbk + n];
        }
    }
    bf16x8_t p[3];
    split8_regs(lo, hi, p);
    uint4* dst = image + blk * (X3W_STAGE / 16) + nb * 64 + lane;
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) dst[pc * 8 * 64] = *reinterpret_cast<const uint4*>(&p[pc]);
}

#ifdef X3W_PROFILE
// dev-only build: cycles per phase of the block loop, summed per wave (512 workgroups from X3W_PROF_FIRST on): wait |
// split | barrier | issue | products | whole loop | epilogue
__device__ unsigned long long x3w_prof[512 * 4 * 8];
extern "C" int jova_debug_x3w_prof(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(x3w_prof), sizeof(unsigned long long) * n);
}
#ifndef X3W_PROF_FIRST
#define X3W_PROF_FIRST 1024
#endif
#define X3W_STAMP(var) const unsigned long long var = __builtin_amdgcn_s_memtime()
#define X3W_ADD(k, v) pf[k] += (v)
#else
#define X3W_STAMP(var)
#define X3W_ADD(k, v)
#endif

#define X3W_GLOAD(dst, ptr, off) \
    asm volatile("global_load_dwordx4 %0, %1, off offset:" #off : "=v"(dst) : "v"(ptr) : "memory")

__global__ void __launch_bounds__(256, 2)
gemm_x3w_kernel(X3WArgs xa) {
    __shared__ __attribute__((aligned(16))) unsigned char stages[3 * X3W_STAGE];
    const GemmArgs& g = xa.g;
    const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, kg = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int slot = blockIdx.x;               // workgroups b, b + 8, ... share an XCD: give each XCD a contiguous run of tiles
    {
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, xcd = slot & 7, loc = slot >> 3;
        slot = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int mt, nt;                          // groups of X3W_NG column tiles; inside a group the column tile runs fastest
    {
        const int ng = X3W_NG > 0 ? X3W_NG : xa.ntn;
        const int per_group = xa.ntm * ng, grp = slot / per_group, in = slot - grp * per_group;
        const int width = min(ng, xa.ntn - grp * ng);
        mt = in / width;
        nt = grp * ng + (in - mt * width);
    }
    const int m0 = mt * 256 + 64 * wave, n0 = nt * 128;
    const int nkb = xa.nkb;
    const float* ap0 = g.A + (long long)min(m0 + i, g.M - 1) * g.sam + 8 * kg;
    const float* ap1 = g.A + (long long)min(m0 + 16 + i, g.M - 1) * g.sam + 8 * kg;
    const float* ap2 = g.A + (long long)min(m0 + 32 + i, g.M - 1) * g.sam + 8 * kg;
    const float* ap3 = g.A + (long long)min(m0 + 48 + i, g.M - 1) * g.sam + 8 * kg;
    const unsigned char* bp = xa.image + (long long)nt * nkb * X3W_STAGE + (wave * 6) * 1024 + lane * 16;
    const unsigned st0 = lds_addr(reinterpret_cast<const float*>(stages)) + (unsigned)(wave * 6) * 1024u;

    f32x4 acc[4][8];
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int nb = 0; nb < 8; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // half 0: row blocks 0 (x) and 1 (y); half 1: row blocks 2 (u) and 3 (v); eight floats of the k32 block each
    f32x4 x0, x1, y0, y1;
    f32x4 u0, u1, v0, v1;
    auto issue_h0 = [&]() {
        X3W_GLOAD(x0, ap0, 0);
        X3W_GLOAD(x1, ap0, 16);
        X3W_GLOAD(y0, ap1, 0);
        X3W_GLOAD(y1, ap1, 16);
    };
    auto issue_h1 = [&]() {
        X3W_GLOAD(u0, ap2, 0);
        X3W_GLOAD(u1, ap2, 16);
        X3W_GLOAD(v0, ap3, 0);
        X3W_GLOAD(v1, ap3, 16);
        ap0 += 32;
        ap1 += 32;
        ap2 += 32;
        ap3 += 32;
    };
    auto issue_b = [&](int stage) {
        const unsigned dst = st0 + (unsigned)stage * X3W_STAGE;
#pragma unroll
        for (int p = 0; p < 6; ++p) dma16(reinterpret_cast<const float*>(bp + p * 1024), dst + p * 1024);
        bp += X3W_STAGE;
    };
    auto split_pair = [&](const f32x4& lo0, const f32x4& hi0, const f32x4& lo1, const f32x4& hi1, bf16x8_t (&pa)[2][3]) {
        const float a[4] = {lo0[0], lo0[1], lo0[2], lo0[3]}, b[4] = {hi0[0], hi0[1], hi0[2], hi0[3]};
        const float c[4] = {lo1[0], lo1[1], lo1[2], lo1[3]}, d[4] = {hi1[0], hi1[1], hi1[2], hi1[3]};
        split8_regs(a, b, pa[0]);
        split8_regs(c, d, pa[1]);
        // (the pieces are complete here: an empty asm that "rewrites" them keeps the split from sinking below the loads)
        asm volatile("" : "+v"(pa[0][0]), "+v"(pa[0][1]), "+v"(pa[0][2]), "+v"(pa[1][0]), "+v"(pa[1][1]), "+v"(pa[1][2])::"memory");
    };
    // the 96 products of one row half: small terms first (A.h B.l | A.m B.m, A.h B.m | A.l B.h, A.m B.h, A.h B.h)
    auto products = [&](const unsigned char* sb, const bf16x8_t (&pa)[2][3], int mb0) {
        bf16x8_t bf[4];
#pragma unroll
        for (int pb = 2; pb >= 0; --pb) {
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
                for (int nb = 0; nb < 4; ++nb)
                    bf[nb] = *reinterpret_cast<const bf16x8_t*>(sb + (pb * 8 + 4 * nh + nb) * 1024);
#pragma unroll
                for (int pa_i = 2 - pb; pa_i >= 0; --pa_i)
#pragma unroll
                    for (int mh = 0; mh < 2; ++mh)
#pragma unroll
                        for (int nb = 0; nb < 4; ++nb)
                            acc[mb0 + mh][4 * nh + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                                bf[nb], pa[mh][pa_i], acc[mb0 + mh][4 * nh + nb], 0, 0, 0);
            }
        }
    };

    issue_b(0);
    if (nkb > 1) issue_b(1);
    issue_h0();
    issue_h1();

    int cur = 0, refill = 2;
#ifdef X3W_PROFILE
    unsigned long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long pf_begin = __builtin_amdgcn_s_memtime();
#endif
    // NEXT: block s + 1 exists (A is requested one block ahead); B2: block s + 2 exists (B two blocks ahead)
    auto block = [&](auto next_tag, auto b2_tag) {
        constexpr bool NEXT = decltype(next_tag)::value, B2 = decltype(b2_tag)::value;
        bf16x8_t pa[2][3];
        X3W_STAMP(p0);
        asm volatile("s_waitcnt vmcnt(4)" ::"v"(x0), "v"(x1), "v"(y0), "v"(y1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        X3W_STAMP(p1);
        split_pair(x0, x1, y0, y1, pa);
        __builtin_amdgcn_sched_barrier(0);
        X3W_STAMP(p2);
        __builtin_amdgcn_s_barrier();    // B(s) of every wave has landed; every wave is past its reads of stage `refill`
        asm volatile("" ::: "memory");
        X3W_STAMP(p3);
        if (B2) issue_b(refill);
        if (NEXT) issue_h0();
        X3W_STAMP(p4);
        const unsigned char* sb = stages + cur * X3W_STAGE + lane * 16;
        products(sb, pa, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifdef X3W_PROFILE
        asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[0][7]), "+v"(acc[1][7]), "+v"(acc[2][7]), "+v"(acc[3][7]));
#endif
        X3W_STAMP(p5);
        if (B2) asm volatile("s_waitcnt vmcnt(10)" ::"v"(u0), "v"(u1), "v"(v0), "v"(v1) : "memory");
        else if (NEXT) asm volatile("s_waitcnt vmcnt(4)" ::"v"(u0), "v"(u1), "v"(v0), "v"(v1) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::"v"(u0), "v"(u1), "v"(v0), "v"(v1) : "memory");
        __builtin_amdgcn_sched_barrier(0);
        X3W_STAMP(p6);
        split_pair(u0, u1, v0, v1, pa);
        __builtin_amdgcn_sched_barrier(0);
        X3W_STAMP(p7);
        if (NEXT) issue_h1();
        X3W_STAMP(p8);
        products(sb, pa, 2);
        cur = cur == 2 ? 0 : cur + 1;
        refill = refill == 2 ? 0 : refill + 1;
#ifdef X3W_PROFILE
        asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[1][0]), "+v"(acc[2][0]), "+v"(acc[3][0]), "+v"(acc[0][7]), "+v"(acc[1][7]), "+v"(acc[2][7]), "+v"(acc[3][7]));
        const unsigned long long p9 = __builtin_amdgcn_s_memtime();
        X3W_ADD(0, (p1 - p0) + (p6 - p5)); X3W_ADD(1, (p2 - p1) + (p7 - p6)); X3W_ADD(2, p3 - p2); X3W_ADD(3, (p4 - p3) + (p8 - p7));
        X3W_ADD(4, (p5 - p4) + (p9 - p8));
#endif
    };
    for (int s = 0; s + 2 < nkb; ++s) block(std::true_type(), std::true_type());
    if (nkb > 1) block(std::true_type(), std::false_type());
    block(std::false_type(), std::false_type());

#ifdef X3W_PROFILE
    const unsigned long long pf_loop_end = __builtin_amdgcn_s_memtime();
#endif
    // epilogue: acc[mb][nb] holds row m0 + 16 mb + i, columns n0 + 16 nb + 4 kg + (0..3): four consecutive columns.
    // Stored as they lie, an instruction would write 32 bytes to each of 32 rows (stamped on the 32x32x16 form: 55 000
    // cycles per tile, a third of the tile's time); each wave turns a 32 x 128 block round in its own 16 KiB of the (now
    // idle) stage memory instead -- 16-byte chunk c of row r at chunk c ^ sw(r), conflict-free for the ds_write_b128
    // down a column and the ds_read_b128 along rows -- and stores whole 128-byte row segments (47 000 cycles: what is
    // left is the store path itself, about 7 bytes per cycle and CU; without its stores the launch takes 285 us
    // instead of 360).
    __syncthreads();                     // every wave is past its last reads of the stages
    float* const reg = reinterpret_cast<float*>(stages) + wave * 4096;
    auto sw = [](int r) { return (r & 7) ^ (((r >> 1) & 1) << 3); };
    auto to_lds = [&](int rb) {
#pragma unroll
        for (int mh = 0; mh < 2; ++mh)
#pragma unroll
            for (int nb = 0; nb < 8; ++nb) {
                const int r = 16 * mh + i, c = 4 * nb + kg;
                *reinterpret_cast<f32x4*>(reg + (r * 32 + (c ^ sw(r))) * 4) = acc[2 * rb + mh][nb];
            }
    };
    // Straight-line form (16-byte accesses): the bias and the residual / mask values of a block are requested BEFORE its
    // stores and nothing is loaded between two stores.  Requests retire in order, so a load behind a store waits for the
    // store's acknowledgement: with a bias load in front of every store (the first version) the 32 stores of a tile ran
    // one round trip each, 47 000 cycles per tile.
    auto epilogue_vec = [&](auto bias_tag, auto r_tag) {
        constexpr bool HB = decltype(bias_tag)::value, HR = decltype(r_tag)::value;
        const int c0 = lane & 7, r0 = lane >> 3;
        f32x4 bs[4], rv[8];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            bs[k] = HB ? *reinterpret_cast<const f32x4*>(g.bias + n0 + 4 * (c0 + 8 * k)) : (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load_r = [&](int rb, int jh) {              // the values of two row groups (8 of the block's 16 chunks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = min(m0 + 32 * rb + r0 + 8 * (2 * jh + j), g.M - 1);
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    rv[4 * j + k] = *reinterpret_cast<const f32x4*>(g.R + (long long)m * g.ldr + n0 + 4 * (c0 + 8 * k));
            }
        };
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            to_lds(rb);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (HR && (j & 1) == 0) load_r(rb, j >> 1);      // (waits for the stores before it: four drains per tile, not 32)
                const int r = r0 + 8 * j, m = m0 + 32 * rb + r;
                f32x4 o[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = c0 + 8 * k;
                    const f32x4 t = *reinterpret_cast<const f32x4*>(reg + (r * 32 + (c ^ sw(r))) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float v = g.alpha * t[e] + bs[k][e];
                        if (HR) v = g.relu == 2 ? (rv[4 * (j & 1) + k][e] > 0.f ? v : 0.f) : v + rv[4 * (j & 1) + k][e];
                        o[k][e] = g.relu == 1 ? fmaxf(v, 0.f) : v;
                    }
                }
                if (m < g.M) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)      // (non-temporal: 232 MB of output pass through, FFN linear1 279 -> 274 us)
                        __builtin_nontemporal_store(o[k], reinterpret_cast<f32x4*>(g.C + (long long)m * g.ldc + n0 + 4 * (c0 + 8 * k)));
                }
            }
        }
    };
    if (xa.vec_c) {
        if (g.bias && g.R) epilogue_vec(std::true_type(), std::true_type());
        else if (g.bias) epilogue_vec(std::true_type(), std::false_type());
        else if (g.R) epilogue_vec(std::false_type(), std::true_type());
        else epilogue_vec(std::false_type(), std::false_type());
    } else {                              // C, residual or bias off a 16-byte boundary: element by element
#pragma unroll                            // (rb must be a constant: a runtime index would put the accumulators in scratch)
        for (int rb = 0; rb < 2; ++rb) {
            to_lds(rb);
#pragma unroll 1
            for (int jk = 0; jk < 16; ++jk) {
                const int j = jk >> 2, k = jk & 3;
                const int r = (lane >> 3) + 8 * j, m = m0 + 32 * rb + r, c = (lane & 7) + 8 * k, n = n0 + 4 * c;
                const f32x4 t = *reinterpret_cast<const f32x4*>(reg + (r * 32 + (c ^ sw(r))) * 4);
                if (m >= g.M) continue;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float v = g.alpha * t[e] + (g.bias ? g.bias[n + e] : 0.f);
                    const float rr = g.R ? g.R[(long long)m * g.ldr + n + e] : 0.f;
                    if (g.relu == 2) v = rr > 0.f ? v : 0.f;
                    else v += rr;
                    if (g.relu == 1) v = fmaxf(v, 0.f);
                    g.C[(long long)m * g.ldc + n + e] = v;
                }
            }
        }
    }
#ifdef X3W_PROFILE
    if (lane == 0 && blockIdx.x >= X3W_PROF_FIRST && blockIdx.x < X3W_PROF_FIRST + 512) {
        unsigned long long* d = x3w_prof + ((size_t)(blockIdx.x - X3W_PROF_FIRST) * 4 + wave) * 8;
        for (int k = 0; k < 5; ++k) d[k] = pf[k];
        d[5] = pf_loop_end - pf_begin;
        d[6] = __builtin_amdgcn_s_memtime() - pf_loop_end;
        d[7] = pf_begin;
    }
#endif
}
#undef X3W_GLOAD


static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// backward products on the bf16 matrix pipe (gemm_bf16x3_kernel): -1 = not decided yet (environment), 0 / 1
static int g_bf16x3 = -1;
// 0 = every product on the fp32 MFMA kernels, 1 = backward products (dX, dW) on bf16 pieces, 2 = forward products too
static int bf16x3_state() {
    if (g_bf16x3 < 0)
        g_bf16x3 = getenv("JOVA_GEMM_NO_BF16X3") ? 0 : (getenv("JOVA_GEMM_BF16X3_BWD_ONLY") ? 1 : 2);
    return g_bf16x3;
}
extern "C" void jova_gemm_set_bf16x3(int mode) { g_bf16x3 = mode <= 0 ? 0 : (mode >= 2 ? 2 : 1); }
extern "C" int jova_gemm_get_bf16x3(void) { return bf16x3_state(); }

// csum / csum_done: the column sums of A, if the form that runs can add them up on the way (aligned form, A row-contiguous)
static int gemm_impl(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                     float* C, int64_t ldc, int M, int N, int K, float alpha, const float* bias,
                     const float* residual, int64_t ldr, int relu, int splitk, int nb1, int nb2, int64_t sA1,
                     int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, void* stream, float* csum,
                     bool* csum_done, const void* b_image = nullptr, size_t* b_image_bytes_query = nullptr) {
    if (csum_done) *csum_done = false;
    if (b_image_bytes_query) *b_image_bytes_query = 0;
    JOVA_CHECK_ARG(M >= 0 && N >= 0 && K >= 0 && nb1 >= 1 && nb2 >= 1, "gemm: bad shape M=%d N=%d K=%d", M, N, K);
    if (M == 0 || N == 0) return JOVA_OK;                    // (an empty product: pointers of empty tensors are null)
    JOVA_CHECK_ARG(A && B && (C || b_image_bytes_query), "gemm: null pointer");
    JOVA_CHECK_ARG(sam == 1 || sak == 1, "gemm: A needs a unit stride (sam=%lld sak=%lld)", (long long)sam, (long long)sak);
    JOVA_CHECK_ARG(sbk == 1 || sbn == 1, "gemm: B needs a unit stride (sbk=%lld sbn=%lld)", (long long)sbk, (long long)sbn);
    JOVA_CHECK_ARG(splitk >= 1 && (splitk == 1 || (!residual && !relu)), "gemm: split-K excludes residual/relu");
    JOVA_CHECK_ARG(relu >= 0 && relu <= 2 && (relu != 2 || residual), "gemm: relu = 2 (mask) needs the mask source as residual");
    if (M == 0 || N == 0) return JOVA_OK;
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.R = residual;
    g.M = M; g.N = N; g.K = K;
    g.sam = sam; g.sak = sak; g.sbk = sbk; g.sbn = sbn; g.ldc = ldc; g.ldr = ldr;
    g.nb2 = nb2; g.sA1 = sA1; g.sA2 = sA2; g.sB1 = sB1; g.sB2 = sB2; g.sC1 = sC1; g.sC2 = sC2;
    g.alpha = alpha; g.relu = relu; g.splitk = splitk;
    g.csum = nullptr;
    const bool a_kc = (sak == 1), b_kc = (sbk == 1);
    const int64_t a_ld = a_kc ? sam : sak, b_ld = b_kc ? sbn : sbk;
    // 16-byte loads need aligned quads, and a quad that straddles the end of the contiguous extent (K for a
    // K-contiguous operand, M / N for a row-contiguous one) must still lie inside the row: the extent is a multiple of
    // 4, or the leading dimension covers the extent rounded up to 4 (padded rows, e.g. the attention probabilities
    // [S, S4]); the elements past the extent are dropped after the load, whatever they hold
    const int a_ext = a_kc ? K : M, b_ext = b_kc ? K : N;
    const bool va = aligned16(A) && (a_ld % 4 == 0) && (sA1 % 4 == 0) && (sA2 % 4 == 0) &&
                    (a_ext % 4 == 0 || a_ld >= ((a_ext + 3) & ~3));
    const bool vb = aligned16(B) && (b_ld % 4 == 0) && (sB1 % 4 == 0) && (sB2 % 4 == 0) &&
                    (b_ext % 4 == 0 || b_ld >= ((b_ext + 3) & ~3));
    g.vecA = va; g.vecB = vb;
    hipStream_t s = (hipStream_t)stream;
    static int fast_env = -1, trace_env = -1;
    if (fast_env < 0) fast_env = getenv("JOVA_GEMM_NO_FAST") ? 0 : 1;
    if (trace_env < 0) trace_env = getenv("JOVA_GEMM_TRACE") ? 1 : 0;                // one line per call on stderr
    if (trace_env)
        fprintf(stderr, "gemm M=%d N=%d K=%d a_kc=%d b_kc=%d va=%d vb=%d nb=%dx%d splitk=%d relu=%d res=%d lda=%lld ldb=%lld ldc=%lld\n",
                M, N, K, (int)a_kc, (int)b_kc, (int)va, (int)vb, nb1, nb2, splitk, relu, residual != nullptr,
                (long long)a_ld, (long long)b_ld, (long long)ldc);
    // backward products (B row-contiguous: dX = dY W, dW = dY^T X) of at least a GFLOP on aligned shapes: fp32 from three
    // bf16 pieces on the bf16 matrix pipe (gemm_bf16x3_kernel).  One launch each, kernel trace, fp32 MFMA kernel -> this one:
    // dX / dW of 30 848 x 2048 x 512: 568 / 533 -> 446 / 417 us; x 512 x 2048: 591 / 527 -> 468 / 406; x 1536 x 512: 424 / 416
    // -> 332 / 324; x 512 x 512: 151 / 153 -> 128 / 123; 23 168 x 256 x 256: 43 / 46 -> 41 / 40 (tools/x3_shapes.py).
    // JOVA_GEMM_NO_BF16X3 keeps everything on the fp32 MFMA kernel.
    static int x3_all = -1;
    if (x3_all < 0) x3_all = getenv("JOVA_GEMM_BF16X3_ALL") ? 1 : 0;              // (measurements: every aligned backward shape)
    const int x3_env = bf16x3_state();
    // forward products (both operands K-contiguous) as well: admitted in round 3 on the evidence of
    // tests/golden/stage_noise_davis_b128.npz (the reference's own fp32 arithmetic with its summation order changed moves
    // the same downstream gradient by up to 9e-4; this form by 1.2e-4), DESIGN.md section 4
    const bool x3_fwd = x3_env >= 2;
    static int x3_old = -1;
    if (x3_old < 0) x3_old = getenv("JOVA_GEMM_BF16X3_OLD") ? 1 : 0;              // (A/B: the register-staged kernel)
    const bool x3_want = x3_env && fast_env && nb1 == 1 && nb2 == 1 && (!b_kc || (x3_fwd && a_kc)) && M % 128 == 0 && N % 128 == 0 &&
                         K % XBK == 0 && K > 0 && va && vb && (x3_all || (long long)M * N * K >= (1ll << 30)) &&
                         (splitk == 1 || ldc == N);
    // A K-contiguous, B a matrix every row tile shares (forward products and data gradients of the linear layers), whole
    // tiles: B split beforehand into the caller's image, A split once per element (gemm_x3w_kernel)
    static int x3w_env = -1;
    if (x3w_env < 0) x3w_env = getenv("JOVA_GEMM_NO_X3W") ? 0 : 1;
    const bool x3w = x3_want && x3w_env && a_kc && splitk == 1 && !x3_old && (long long)N * K * 6 < (1ll << 31);
    if (b_image_bytes_query) {
        *b_image_bytes_query = x3w ? (size_t)N * K * 6 : 0;
        return JOVA_OK;
    }
    if (x3w && b_image) {
        JOVA_CHECK_ARG(aligned16(b_image), "gemm: the B image must be 16-byte aligned");
        X3WArgs xw;
        xw.g = g;
        xw.image = (const unsigned char*)b_image;
        xw.ntm = cdiv(M, 256);
        xw.ntn = N / 128;
        xw.nkb = K / 32;
        xw.vec_c = aligned16(C) && ldc % 4 == 0 && (!residual || (aligned16(residual) && ldr % 4 == 0)) &&
                   (!bias || aligned16(bias));
        const long long nwg = (long long)xw.ntm * xw.ntn;
        JOVA_CHECK_ARG(nwg < (1ll << 31), "gemm: too many tiles");
        hipLaunchKernelGGL(gemm_x3w_kernel, dim3((unsigned)nwg), dim3(256), 0, s, xw);
        JOVA_CHECK_LAUNCH("gemm_x3w");
        return JOVA_OK;
    }
    // dW (both operands row-contiguous) on the register-staged kernel (262-290 us against 292-332 us), dX (A K-contiguous)
    // on the LDS-DMA form of the persistent kernel (338 us against 364 us on average): kernel trace of the Davis step
    // weight gradients with at least eight 256 x 256 tiles (the FFN and QKV layers): the big-tile form, the contraction
    // split over about one workgroup per CU
    static int x3_dw_big = -1;
    if (x3_dw_big < 0) x3_dw_big = getenv("JOVA_GEMM_NO_DW256") ? 0 : 1;
    if (x3_want && x3_dw_big && !x3_old && !a_kc && !b_kc && splitk > 1 && !relu && M % 256 == 0 && N % 256 == 0 &&
        (long long)(M / 256) * (N / 256) >= 8 && (long long)(M / 256) * (N / 256) <= 256 && K / XBK >= 64) {
        X3Args xa;
        xa.g = g;
        xa.ntm = M / 256;
        xa.ntn = N / 256;
        xa.nk_total = K / XBK;
        const int tiles = xa.ntm * xa.ntn;
        int parts = 256 / tiles;
        if (parts > xa.nk_total / 8) parts = xa.nk_total / 8;            // at least 8 k32 tiles per part
        if (parts < 1) parts = 1;
        xa.per = (xa.nk_total + parts - 1) / parts;
        xa.parts = (xa.nk_total + xa.per - 1) / xa.per;
        if (csum) {
            xa.g.csum = csum;
            if (csum_done) *csum_done = true;
        }
        hipLaunchKernelGGL(gemm_bf16x3_dw_kernel, dim3((unsigned)(tiles * xa.parts)), dim3(512), 0, s, xa);
        JOVA_CHECK_LAUNCH("gemm_bf16x3_dw");
        return JOVA_OK;
    }
    if (x3_want && (x3_old || !a_kc)) {
        X3Args xa;
        xa.g = g;
        xa.ntm = M / 128;
        xa.ntn = N / 128;
        xa.nk_total = K / XBK;
        xa.per = (xa.nk_total + splitk - 1) / splitk;
        xa.parts = (xa.nk_total + xa.per - 1) / xa.per;
        if (csum && !a_kc) {
            xa.g.csum = csum;
            if (csum_done) *csum_done = true;
        }
        const long long nwg = (long long)xa.ntm * xa.ntn * xa.parts;
        JOVA_CHECK_ARG(nwg < (1ll << 31), "gemm: too many tiles");
        if (a_kc && b_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, true>), dim3((unsigned)nwg), dim3(256), 0, s, xa);
        else if (a_kc) hipLaunchKernelGGL((gemm_bf16x3_kernel<true, false>), dim3((unsigned)nwg), dim3(256), 0, s, xa);
        else hipLaunchKernelGGL((gemm_bf16x3_kernel<false, false>), dim3((unsigned)nwg), dim3(256), 0, s, xa);
        JOVA_CHECK_LAUNCH("gemm_bf16x3");
        return JOVA_OK;
    }
    const bool fast = fast_env && nb1 == 1 && nb2 == 1 && M % 128 == 0 && N % 128 == 0 && K % FBK == 0 && K > 0 && va &&
                      vb && aligned16(C) && (!residual || true);
    // ragged / batched aligned form: both operands K-contiguous, no residual / ReLU / split (S <= 256 of the attention
    // products; S = 279 would pad to 384 and stays on the 96-tiles of the generic form)
    static int rag_env = -1;
    if (rag_env < 0) rag_env = getenv("JOVA_GEMM_NO_RAGGED") ? 0 : 1;
    if (!fast && fast_env && rag_env && a_kc && b_kc && K % FBK == 0 && K > 0 && va && vb && !residual && !relu && splitk == 1 &&
        (nb1 > 1 || nb2 > 1)) {
        const long long tm = cdiv(M, 128), tn = cdiv(N, 128), nbt = (long long)nb1 * nb2;
        // padded work against the generic form's best tile size (same weights as its own choice below), this form
        // running at about 1.15 x the generic 128-tile rate (measured at S = 241: 105 against 122 us).  Only up to 2 x 2
        // tiles per matrix: at S = 400-1100 (Metz + PSC, long-protein batches) the generic form measured 1-2 % better on
        // the step -- with K = 256 an item is 16 k-tiles long and this kernel's per-item costs weigh more than its rate
        double generic = 1e300;
        for (int w = 4; w >= 2; --w) {
            const double c = (double)cdiv(M, 32 * w) * cdiv(N, 32 * w) * (32 * w) * (32 * w) * (w == 4 ? 1.0 : (w == 3 ? 1.2 : 1.25));
            if (c < generic) generic = c;
        }
        if (tm <= 2 && tn <= 2 && (double)tm * tn * 128 * 128 <= 1.15 * generic && tm * tn * nbt >= 256 && tm * tn * nbt < (1ll << 30)) {
            FastArgs fa;
            fa.g = g;
            fa.ntm = (int)tm;
            fa.ntn = (int)tn;
            fa.tpb = (int)(tm * tn);
            fa.nk_total = K / FBK;
            fa.full = fa.items = (int)(tm * tn * nbt);
            fa.rem = 0;
            fa.per = fa.nk_total;
            const int slots = 256 * FWGS, nwg = fa.items < slots ? fa.items : slots;
            hipLaunchKernelGGL((gemm_f32_fast_kernel<true, true, true>), dim3(nwg), dim3(256), 0, s, fa);
            JOVA_CHECK_LAUNCH("gemm_f32_fast (ragged)");
            return JOVA_OK;
        }
    }
    if (fast) {
        if (csum && !a_kc) {
            g.csum = csum;
            if (csum_done) *csum_done = true;
        }
        FastArgs fa;
        fa.g = g;
        fa.tpb = 0;
        fa.ntm = M / 128;
        fa.ntn = N / 128;
        fa.nk_total = K / FBK;
        const int tiles = fa.ntm * fa.ntn, slots = 256 * FWGS;       // persistent workgroups, FWGS per CU
        if (splitk > 1) {                                            // explicit split-K: C accumulates
            fa.per = (fa.nk_total + splitk - 1) / splitk;
            fa.full = 0;
            fa.rem = tiles;
            fa.items = tiles * ((fa.nk_total + fa.per - 1) / fa.per);    // splits that own at least one k-tile
        } else {
            fa.full = tiles;
            fa.rem = 0;
            fa.per = fa.nk_total;
            fa.items = tiles;
            // ragged last round: split its tiles along K over the idle workgroups (needs C rows zeroed first, so C
            // must not double as the residual, be dense, and ReLU cannot apply to partial sums)
            const int rem = tiles % slots;
            const int parts = rem > 0 ? slots / rem : 1;
            const int max_parts = fa.nk_total / 8;                   // at least 8 k-tiles per part
            const int use = parts < max_parts ? parts : max_parts;
            // a lone product of a few tiles with a long K (the ecfp view projection: 128 x 512 x 1024 is four tiles):
            // all of it is split along K, partial sums added onto the zeroed C, the bias rides with part 0
            int few = fa.nk_total / 8;
            if (few > slots / (tiles > 0 ? tiles : 1)) few = slots / (tiles > 0 ? tiles : 1);
            if (tiles * 8 <= slots && few >= 2 && !relu && !residual && ldc == N) {
                fa.full = 0;
                fa.rem = tiles;
                fa.per = (fa.nk_total + few - 1) / few;
                fa.items = tiles * ((fa.nk_total + fa.per - 1) / fa.per);
                JOVA_ZERO(C, (size_t)M * ldc * sizeof(float), s);
            } else if (tiles > slots && use >= 2 && !relu && residual != C && ldc == N) {
                fa.full = tiles - rem;
                fa.rem = rem;
                fa.per = (fa.nk_total + use - 1) / use;
                fa.items = fa.full + rem * ((fa.nk_total + fa.per - 1) / fa.per);
                const long long row0 = (long long)(fa.full / fa.ntn) * 128;
                JOVA_ZERO(C + row0 * ldc, (size_t)(M - row0) * ldc * sizeof(float), s);
            }
        }
        const int nwg = fa.items < slots ? fa.items : slots;
#define GEMM_FAST(AK, BK_)                                                                                          \
    do {                                                                                                            \
        if (x3_want) hipLaunchKernelGGL((gemm_f32_fast_kernel<AK, BK_, false, true>), dim3(nwg), dim3(256), 0, s, fa);  \
        else hipLaunchKernelGGL((gemm_f32_fast_kernel<AK, BK_>), dim3(nwg), dim3(256), 0, s, fa);                   \
    } while (0)
        if (a_kc && b_kc) GEMM_FAST(true, true);
        else if (a_kc && !b_kc) GEMM_FAST(true, false);
        else if (!a_kc && b_kc) GEMM_FAST(false, true);
        else GEMM_FAST(false, false);
#undef GEMM_FAST
        JOVA_CHECK_LAUNCH("gemm_f32_fast");
        return JOVA_OK;
    }
    long long gz = (long long)nb1 * nb2 * splitk;
    JOVA_CHECK_ARG(gz <= 65535, "gemm: batch*splitk %lld exceeds grid.z", gz);
    // tile size: least padded work (smaller tiles pay a little more staging per flop), then enough workgroups for
    // every CU; JOVA_GEMM_W pins it (tests)
    static int w_env = -1;
    if (w_env < 0) {
        const char* e = getenv("JOVA_GEMM_W");
        w_env = e ? atoi(e) : 0;
    }
    int Wsel = 4;
    if (w_env >= 2 && w_env <= 4) Wsel = w_env;
    else {
        double best = 1e300;
        for (int w = 4; w >= 2; --w) {
            const long long ts = 32 * w;
            const long long wgs = (long long)cdiv(M, ts) * cdiv(N, ts) * gz;
            // time estimate in units of one SIMD's work: all 1024 SIMDs busy, or one wave's share of a tile
            const double total = (double)wgs * ts * ts / 1024.0, alone = (double)ts * ts / w;
            // (measured on the per-(b,h) attention products at S = 279: 96-tiles run at ~0.85 of the 128-tile rate)
            const double cost = (total > alone ? total : alone) * (w == 4 ? 1.0 : (w == 3 ? 1.2 : 1.25));
            if (cost < best) { best = cost; Wsel = w; }
        }
    }
    // long batched contractions (attention products of long-protein batches): folded accumulation chains, 128-tiles
    const bool fold = K > 512 && (long long)nb1 * nb2 > 1 && splitk == 1 && va && vb;
    if (fold) Wsel = 4;
    const long long ts = 32 * Wsel;
    // a lone small product with a long K (the FC head at M = 128): split K over otherwise idle CUs, partial sums
    // added onto a zeroed C (bias comes with split 0; ReLU / residual cannot apply to partial sums)
    static int skt_env = -1, skmin_env = -1;
    // (targets measured on the thin products of the Davis step, tools/bench_small_gemm.py under a kernel trace: with
    // 4096 waves / K >= 128 the data gradients of the 194- and 22-wide view projections go from 55.7 / 53.3 to 37.7 /
    // 29.3 us; 1024 / 256 were the round-2 values)
    if (skt_env < 0) skt_env = getenv("JOVA_GEMM_SKT") ? atoi(getenv("JOVA_GEMM_SKT")) : 4096;
    if (skmin_env < 0) skmin_env = getenv("JOVA_GEMM_SKMINK") ? atoi(getenv("JOVA_GEMM_SKMINK")) : 128;
    if (splitk == 1 && gz == 1 && !relu && !residual && K >= skmin_env) {
        const long long waves = (long long)cdiv(M, ts) * cdiv(N, ts) * Wsel;
        int sk = (int)(skt_env / (waves > 0 ? waves : 1));
        if (sk > K / 64) sk = K / 64;
        if (sk > 32) sk = 32;
        if (sk >= 2) {
            {
                const int rcz = jova_zero_fill(C, (size_t)N * sizeof(float), (size_t)M, (size_t)ldc * sizeof(float), s);
                if (rcz != JOVA_OK) return rcz;
            }
            g.splitk = sk;
            gz = sk;
        }
    }
    dim3 grid((unsigned)((long long)cdiv(M, ts) * cdiv(N, ts)), 1, (unsigned)gz);
#define GEMM_LAUNCH(AK, BK_, VA_, VB_, W_) \
    hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, VA_, VB_, W_>), grid, dim3(64 * W_), 0, s, g)
#define GEMM_W(AK, BK_, VA_, VB_)                                              \
    do {                                                                       \
        if (Wsel == 4) GEMM_LAUNCH(AK, BK_, VA_, VB_, 4);                      \
        else if (Wsel == 3) GEMM_LAUNCH(AK, BK_, VA_, VB_, 3);                 \
        else GEMM_LAUNCH(AK, BK_, VA_, VB_, 2);                                \
    } while (0)
#define GEMM_VEC(AK, BK_)                                                     \
    do {                                                                      \
        if (va && vb) GEMM_W(AK, BK_, true, true);                            \
        else if (va) GEMM_W(AK, BK_, true, false);                            \
        else if (vb) GEMM_W(AK, BK_, false, true);                            \
        else GEMM_W(AK, BK_, false, false);                                   \
    } while (0)
#define GEMM_FOLDED(AK, BK_) hipLaunchKernelGGL((gemm_f32_kernel<AK, BK_, true, true, 4, true>), grid, dim3(256), 0, s, g)
    if (fold) {
        if (a_kc && b_kc) GEMM_FOLDED(true, true);
        else if (a_kc && !b_kc) GEMM_FOLDED(true, false);
        else if (!a_kc && b_kc) GEMM_FOLDED(false, true);
        else GEMM_FOLDED(false, false);
    } else if (a_kc && b_kc) GEMM_VEC(true, true);
    else if (a_kc && !b_kc) GEMM_VEC(true, false);
    else if (!a_kc && b_kc) GEMM_VEC(false, true);
    else GEMM_VEC(false, false);
    JOVA_CHECK_LAUNCH("gemm_f32");
    return JOVA_OK;
}

extern "C" int jova_gemm_f32(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                             float* C, int64_t ldc, int M, int N, int K, float alpha, const float* bias,
                             const float* residual, int64_t ldr, int relu, int splitk, int nb1, int nb2, int64_t sA1,
                             int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2, void* stream) {
    return gemm_impl(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, bias, residual, ldr, relu, splitk, nb1, nb2, sA1,
                     sA2, sB1, sB2, sC1, sC2, stream, nullptr, nullptr);
}

// The same product with B split beforehand (see gemm_x3w_kernel).  jova_gemm_b_image_bytes: the size of the image this
// product would use -- 0 when it would not run on that form (shape, alignment, mode), in which case jova_gemm_f32_img
// ignores the image.  jova_gemm_b_image writes the image of B(k, n) = B[k sbk + n sbn], N % 128 == 0, K % 32 == 0.
extern "C" size_t jova_gemm_b_image_bytes(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk,
                                          int64_t sbn, int M, int N, int K, int nb1, int nb2, int splitk) {
    size_t need = 0;
    if (!A || !B || M <= 0 || N <= 0 || K <= 0 || nb1 < 1 || nb2 < 1 || splitk < 1 || (sam != 1 && sak != 1) ||
        (sbk != 1 && sbn != 1))
        return 0;
    const int rc = gemm_impl(A, sam, sak, B, sbk, sbn, nullptr, N, M, N, K, 1.f, nullptr, nullptr, 0, 0, splitk, nb1, nb2, 0,
                             0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, &need);
    return rc == JOVA_OK ? need : 0;
}

extern "C" int jova_gemm_b_image(const float* B, int64_t sbk, int64_t sbn, int N, int K, void* image, void* stream) {
    JOVA_CHECK_ARG(B && image && N > 0 && K > 0 && N % 128 == 0 && K % 32 == 0, "gemm_b_image: bad argument");
    JOVA_CHECK_ARG((sbk == 1 && sbn % 4 == 0 && aligned16(B)) || (sbn == 1 && sbk >= N),
                   "gemm_b_image: B needs a unit stride (and 16-byte aligned rows when K-contiguous)");
    JOVA_CHECK_ARG(aligned16(image) && (long long)N * K * 6 < (1ll << 31), "gemm_b_image: image misaligned or too large");
    const long long units = (long long)N * K / 8;
    hipLaunchKernelGGL(gemm_b_image_kernel, dim3((unsigned)cdiv(units, 256)), dim3(256), 0, (hipStream_t)stream, B,
                       (long long)sbk, (long long)sbn, N, K, (uint4*)image);
    JOVA_CHECK_LAUNCH("gemm_b_image");
    return JOVA_OK;
}

extern "C" int jova_gemm_f32_img(const float* A, int64_t sam, int64_t sak, const float* B, int64_t sbk, int64_t sbn,
                                 float* C, int64_t ldc, int M, int N, int K, float alpha, const float* bias,
                                 const float* residual, int64_t ldr, int relu, int splitk, int nb1, int nb2, int64_t sA1,
                                 int64_t sA2, int64_t sB1, int64_t sB2, int64_t sC1, int64_t sC2,
                                 const void* b_image_or_null, void* stream) {
    return gemm_impl(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, bias, residual, ldr, relu, splitk, nb1, nb2, sA1,
                     sA2, sB1, sB2, sC1, sC2, stream, nullptr, nullptr, b_image_or_null, nullptr);
}

// dw[n_out, k_in] += dy[rows, n_out]^T x[rows, k_in];  db[n_out] += column sums of dy  (the two gradients of an nn.Linear).
// On the aligned form the column sums are added up from the A fragments the product reads anyway; otherwise a
// separate pass over dy does it.
extern "C" int jova_linear_bwd_weight(const float* dy, int64_t ld_dy, const float* x, int64_t ld_x, int64_t rows, int n_out,
                                      int k_in, float* dw, float* db_or_null, int splitk, void* stream) {
    JOVA_CHECK_ARG(dy && x && dw && rows >= 0 && n_out >= 1 && k_in >= 1 && ld_dy >= n_out && ld_x >= k_in && splitk >= 1 &&
                   rows < (1ll << 31), "linear_bwd_weight: bad argument");
    if (rows == 0) return JOVA_OK;
    bool done = false;
    int rc;
    if (splitk > 1)
        rc = gemm_impl(dy, 1, ld_dy, x, ld_x, 1, dw, k_in, n_out, k_in, (int)rows, 1.f, nullptr, nullptr, 0, 0, splitk, 1, 1,
                       0, 0, 0, 0, 0, 0, stream, db_or_null, &done);
    else
        rc = gemm_impl(dy, 1, ld_dy, x, ld_x, 1, dw, k_in, n_out, k_in, (int)rows, 1.f, nullptr, dw, k_in, 0, 1, 1, 1, 0, 0,
                       0, 0, 0, 0, stream, db_or_null, &done);
    if (rc != JOVA_OK) return rc;
    if (db_or_null && !done) return jova_colsum_add(dy, rows, n_out, ld_dy, db_or_null, stream);
    return JOVA_OK;
}
