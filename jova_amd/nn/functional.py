"""autograd glue between the reference-shaped nn.Modules and the HIP kernels (C ABI in include/jova_hip.h).

Every Function's forward/backward only (a) allocates outputs with torch (device memory is PyTorch's job here) and
(b) enqueues kernels of libjova_hip.so on torch's current stream.  No torch math runs on the hot path.

Parameter gradients: the kernels *accumulate*.  If a parameter already has a ``.grad`` (the trainer keeps all of
them as views of one flat buffer that it zeroes once per step), the kernels add straight into it and the Function
returns ``None`` for that input; otherwise a zero tensor is allocated, filled and returned to autograd.
"""
import ctypes
import math

import numpy as np
import torch

from .. import _lib
from .._lib import P, call, ptr_array, stream

_F32 = torch.float32


def _empty(shape, like, dtype=_F32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _zeros(shape, like, dtype=_F32):
    """Zero-filled device tensor; the fill is the library's own kernel on the stream (no torch math on the hot path)."""
    t = torch.empty(shape, dtype=dtype, device=like.device)
    if t.is_cuda and dtype == _F32 and t.numel():
        call("jova_zero", P(t), t.numel(), stream())
        return t
    return t.zero_()


_SEG_OFF = {}


def _seg_off_device(seg_off, device):
    """int32 device copy of a view-offset list, made once per distinct (list, device): JointAttention runs every step with
    one of a handful of (1, A_max, P_max) combinations, and a pageable host->device copy per call would sit in the stream
    (and could not be captured into a hipGraph)."""
    key = (tuple(seg_off), device)
    t = _SEG_OFF.get(key)
    if t is None:
        if len(_SEG_OFF) > 4096:
            _SEG_OFF.clear()
        t = _SEG_OFF[key] = torch.tensor(seg_off, dtype=torch.int32).to(device)
    return t


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _atoms_ld(f):
    """Row stride (floats) of the fp32 copy of int64 atom features: whole 64-byte pieces, so the graph-conv kernels
    can copy rows HBM -> LDS in 16-byte chunks without reading past a row."""
    return (f + 15) // 16 * 16


class _GradSink(object):
    """Where a parameter's gradient is accumulated (see module docstring)."""

    def __init__(self, params):
        self.targets, self.returns = [], []
        for p in params:
            if p is None:
                self.targets.append(None)
                self.returns.append(None)
            elif not p.requires_grad:            # frozen parameter: kernels still need somewhere to write
                self.targets.append(torch.zeros_like(p))
                self.returns.append(None)
            elif p.grad is not None:
                self.targets.append(p.grad)
                self.returns.append(None)
            else:
                g = torch.zeros_like(p)
                self.targets.append(g)
                self.returns.append(g)


# ------------------------------------------------------------------------------------------------------ topology
class Topology(object):
    """Device-side view of one batch graph: the [1], [3], [4:] entries of the reference's gconv input list
    (jova/data/data.py:291-296) plus what the kernels derive from them once per batch."""

    def __init__(self, deg_slice, adj_lists, n_atoms_lst, n_atoms, sym_host=None):
        dev = deg_slice.device
        _lib.require_gpu(deg_slice)
        self.deg_slice = _c(deg_slice.to(torch.int64))
        assert self.deg_slice.shape == (11, 2), "deg_slice must be [11, 2] (max_deg 10, min_deg 0)"
        assert len(adj_lists) == 10, "expected adjacency lists for degrees 1..10"
        self.adj = []
        for d, a in enumerate(adj_lists, start=1):
            _lib.require_gpu(a)
            a = _c(a.to(torch.int32))
            assert a.numel() == 0 or (a.dim() == 2 and a.shape[1] == d), "adjacency list %d has shape %s" % (d, a.shape)
            self.adj.append(a)
        self.adj_ptrs = ptr_array(self.adj)
        self.n_atoms = int(n_atoms)
        self.n_atoms_lst = [int(v) for v in n_atoms_lst] if n_atoms_lst is not None else None
        if sym_host is None:
            self.sym_flag = torch.empty(1, dtype=torch.int32, device=dev)
            call("jova_adj_symmetry", self.n_atoms, P(self.deg_slice), self.adj_ptrs, P(self.sym_flag), stream())
        else:
            # the collator checked the lists on the host (jova_collate_gconv): no kernel on a stream that may not have
            # the uploaded lists yet (prefetch thread), and no device->host read later
            self.sym_flag = torch.full((1,), int(bool(sym_host)), dtype=torch.int32, device=dev)
        self._mol_off = None
        self._atoms_f32 = None
        self._sym_host = None if sym_host is None else bool(sym_host)
        self._deg_slice_host = None
        self._rev_pos = None
        self._block_order = None
        self._row_table = None

    def atoms_f32(self, atoms):
        """fp32 copy of the batch's int64 atom features, rows zero-padded to a multiple of 16 floats (made once)."""
        key = (atoms.data_ptr(), tuple(atoms.shape))
        if self._atoms_f32 is None or self._atoms_f32[0] != key:
            n, f = atoms.shape
            ld = _atoms_ld(f)
            out = torch.empty((n, ld), dtype=_F32, device=atoms.device)
            call("jova_atoms_to_f32", P(atoms), n, f, ld, P(out), stream())
            self._atoms_f32 = (key, out, atoms)      # keeps `atoms` alive so the pointer key stays unique
        return self._atoms_f32[1]

    @property
    def rev_pos(self):
        """uint8 per directed edge: position of the source atom in its neighbour's list (symmetric graphs; once per
        batch).  None when the adjacency is not symmetric."""
        if self._rev_pos is None and self.symmetric:
            n_edges = sum(int(a.numel()) for a in self.adj)
            self._rev_pos = torch.empty(max(n_edges, 1), dtype=torch.uint8, device=self.deg_slice.device)
            call("jova_adj_reverse_pos", self.n_atoms, P(self.deg_slice), self.adj_ptrs, P(self._rev_pos), stream())
        return self._rev_pos

    @property
    def block_order(self):
        """XCD-aware order of the 16-row blocks for the gather kernels (jova_block_order; once per batch; the
        collator ships it inside its packed upload)."""
        if self._block_order is None:
            n = (self.n_atoms + 15) // 16
            host = np.zeros(max(n, 1), dtype=np.int32)
            call("jova_block_order", ctypes.c_void_p(self.deg_slice_host.ctypes.data), self.n_atoms,
                 ctypes.c_void_p(host.ctypes.data))
            self._block_order = torch.from_numpy(host).to(self.deg_slice.device)
        return self._block_order

    @property
    def row_table(self):
        """jova_row_table of the batch (once per batch): row, degree, first four neighbours and reverse positions of
        every row in launch order, so the pool kernels start gathering after one read."""
        if self._row_table is None and self.n_atoms:
            n_blocks = (self.n_atoms + 15) // 16
            self._row_table = torch.empty((n_blocks * 16, 8), dtype=torch.int32, device=self.deg_slice.device)
            call("jova_row_table", self.n_atoms, P(self.deg_slice), self.adj_ptrs, P(self.rev_pos),
                 P(self.block_order), P(self._row_table), stream())
        return self._row_table

    @property
    def deg_slice_host(self):
        """Host copy of deg_slice (numpy int64 [11, 2]): handed over by the collator, which builds it on the host;
        otherwise one device->host read per batch, cached."""
        if self._deg_slice_host is None:
            self._deg_slice_host = np.ascontiguousarray(self.deg_slice.cpu().numpy(), dtype=np.int64)
        return self._deg_slice_host

    @property
    def symmetric(self):
        """Host copy of the symmetry flag (one device->host read per batch, cached)."""
        if self._sym_host is None:
            self._sym_host = bool(self.sym_flag.item())
        return self._sym_host

    @property
    def mol_off(self):
        if self._mol_off is None:
            off = [0]
            for n in self.n_atoms_lst:
                off.append(off[-1] + n)
            self._mol_off = torch.tensor(off, dtype=torch.int32).to(self.deg_slice.device, non_blocking=True)
        return self._mol_off


def topology_of(input_data):
    """Topology of a gconv input list, built once per batch and cached on the deg_slice tensor object."""
    ds = input_data[1]
    topo = getattr(ds, "_jova_topology", None)
    if topo is None:
        topo = Topology(ds, input_data[4:], input_data[3], input_data[0].shape[0])
        try:
            ds._jova_topology = topo
        except Exception:
            pass
    return topo


# ------------------------------------------------------------------------------------------------------ graph ops
class GraphConvFunction(torch.autograd.Function):
    """GraphConvLayer.forward (jova/nn/layers.py:295-347): weight re-pack + one launch; backward two launches.

    int64 atom features (layer 1) are converted once per batch to fp32 rows padded to a multiple of 4 floats
    (exact: they are small integers) and cached on the batch topology; the kernels then read whole rows with
    16-byte loads."""

    @staticmethod
    def forward(ctx, atoms, topo, relu, stats, *wb):
        weights, biases = wb[:21], wb[21:]
        atoms = _c(atoms)
        is_i64 = atoms.dtype == torch.int64
        if not is_i64 and atoms.dtype != _F32:
            raise RuntimeError("GraphConvLayer: atom features must be int64 or float32, got %s" % atoms.dtype)
        n, f_in = atoms.shape
        f_out = weights[0].shape[0]
        xf, ld = (topo.atoms_f32(atoms), _atoms_ld(f_in)) if is_i64 else (atoms, f_in)
        out = _empty((n, f_out), xf)
        ws = _empty((int(_lib.lib().jova_graphconv_workspace_floats(f_in, f_out)),), weights[0])
        call("jova_graphconv_fwd", P(xf), 0, ld, n, f_in, f_out, P(topo.deg_slice), topo.adj_ptrs,
             ptr_array(weights), ptr_array(biases), int(relu), P(out), P(stats), P(ws),
             ctypes.c_void_p(topo.deg_slice_host.ctypes.data), stream())
        ctx.topo, ctx.relu, ctx.is_i64, ctx.params, ctx.dims = topo, relu, is_i64, wb, (n, f_in, f_out, ld)
        ctx.save_for_backward(xf, out)
        ctx.weights = weights
        return out

    @staticmethod
    def backward(ctx, dy):
        xf, out = ctx.saved_tensors
        topo = ctx.topo
        dy = _c(dy)
        n, f_in, f_out, ld = ctx.dims
        sink = _GradSink(ctx.params)
        need_dx = (not ctx.is_i64) and ctx.needs_input_grad[0]
        dx = _empty((n, f_in), out) if need_dx else None
        gbuf = _empty((n, f_in), out) if need_dx else None
        dws = _empty((int(_lib.lib().jova_graphconv_bwd_scratch_floats()),), out) if f_in % 32 else None
        # (a BatchNorm that consumes this layer's ReLU output applies the ReLU backward itself: mask_fused)
        mask = ctx.relu and not getattr(ctx, "mask_fused", False)
        call("jova_graphconv_bwd", P(xf), 0, ld, n, f_in, f_out, P(topo.deg_slice), topo.adj_ptrs,
             ptr_array(ctx.weights), P(dy), P(out) if mask else None, ptr_array(sink.targets[:21]),
             ptr_array(sink.targets[21:]), P(dx), P(gbuf), P(topo.sym_flag),
             ctypes.c_void_p(topo.deg_slice_host.ctypes.data), P(dws), P(topo.block_order) if need_dx else None,
             stream())
        return (dx, None, None, None) + tuple(sink.returns)


class GraphPoolFunction(torch.autograd.Function):
    """GraphPool.forward (jova/nn/layers.py:356-393)."""

    @staticmethod
    def forward(ctx, atoms, topo):
        atoms = _c(atoms.float())
        n, f = atoms.shape
        out = _empty((n, f), atoms)
        arg = _empty((n, f), atoms, torch.uint8)
        call("jova_graphpool_fwd", P(atoms), n, f, P(topo.deg_slice), topo.adj_ptrs, P(out), P(arg),
             P(topo.block_order) if n else None, P(topo.row_table) if n else None, stream())
        ctx.topo = topo
        ctx.save_for_backward(arg)
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        topo = ctx.topo
        dy = _c(dy)
        n, f = dy.shape
        dx = _empty((n, f), dy)
        call("jova_graphpool_bwd", P(dy), P(arg), n, f, P(topo.deg_slice), topo.adj_ptrs, P(dx), P(topo.sym_flag),
             int(topo.symmetric), P(topo.rev_pos), P(topo.block_order) if n else None,
             P(topo.row_table) if n else None, stream())
        return dx, None


class Gather2DFunction(torch.autograd.Function):
    """GraphGather2D's split / pad / stack (jova/nn/layers.py:429-432)."""

    @staticmethod
    def forward(ctx, atoms, topo):
        atoms = _c(atoms)
        n, g = atoms.shape
        lst = topo.n_atoms_lst
        if sum(lst) != n:
            raise RuntimeError("split_with_sizes expects n_atoms_lst to sum to %d, got %d" % (n, sum(lst)))
        a_max, bsz = max(lst), len(lst)
        out = _empty((a_max, bsz, g), atoms)
        call("jova_gather2d_fwd", P(atoms), P(topo.mol_off), a_max, bsz, g, P(out), stream())
        ctx.topo, ctx.shape = topo, (n, g, a_max, bsz)
        return out

    @staticmethod
    def backward(ctx, dout):
        n, g, a_max, bsz = ctx.shape
        dout = _c(dout)
        dx = _empty((n, g), dout)
        call("jova_gather2d_bwd", P(dout), P(ctx.topo.mol_off), a_max, bsz, g, P(dx), stream())
        return dx, None


class NonsatFunction(torch.autograd.Function):
    """nonsat_activation (jova/nn/models.py:458-478): data-dependent trip count decided on the device."""

    last_n_eval = None          # device int32 [3] of the most recent call (evaluations, arrivals, timeout flag)

    @classmethod
    def raise_if_timed_out(cls):
        """Reads the most recent call's give-up flag (ONE device->host read: call it off the hot loop, e.g. once per
        epoch as Trainer.check_health does).  The kernel also NaN-fills what it gave up on, so the loss trips first."""
        if cls.last_n_eval is not None and int(cls.last_n_eval[2].item()) != 0:
            raise RuntimeError("jova_nonsat_fwd: the in-kernel barrier gave up (workgroups not co-resident: is another "
                               "process using this GPU?); its outputs were poisoned with NaN")

    @staticmethod
    def forward(ctx, x, ep, max_iter):
        x = _c(x)
        n = x.numel()
        out = torch.empty_like(x)
        dfac = torch.empty_like(x)
        n_eval = _empty((3,), x, torch.int32)
        ws = _empty((int(_lib.lib().jova_nonsat_workspace_bytes(n)),), x, torch.uint8)
        call("jova_nonsat_fwd", P(x), n, float(ep), int(max_iter), P(out), P(dfac), P(n_eval), P(ws), stream())
        NonsatFunction.last_n_eval = n_eval
        ctx.save_for_backward(dfac)
        return out

    @staticmethod
    def backward(ctx, dout):
        (dfac,) = ctx.saved_tensors
        dout = _c(dout)
        dx = torch.empty_like(dout)
        call("jova_mul", P(dout), P(dfac), dout.numel(), P(dx), stream())
        return dx, None, None


# ------------------------------------------------------------------------------------------------------ dense helpers
def _addr(t, offset=0):
    return ctypes.c_void_p(t.data_ptr() + 4 * offset)


PROBE = {}      # tag -> list of (start event, end event, flops): bench.py times one launch shape live (HIP events)


def gemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha=1.0, bias=None, residual=None, ldr=0, relu=False, splitk=1,
         nb1=1, nb2=1, sA=(0, 0), sB=(0, 0), sC=(0, 0), tag=None, _img=None):
    """Thin wrapper over jova_gemm_f32; A, B, C, bias, residual are ctypes addresses (see _addr)."""
    img = _img
    if img is None and nb1 == 1 and nb2 == 1:
        img = _b_image(A, sam, sak, B, sbk, sbn, M, N, K, splitk)          # (outside the probe's events: its own launch)
    if tag is not None and tag in PROBE:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        gemm(A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, alpha, bias, residual, ldr, relu, splitk, nb1, nb2, sA, sB, sC,
             _img=img)
        e1.record()
        PROBE[tag].append((e0, e1, 2.0 * M * N * K * nb1 * nb2))
        return
    if img is not None:
        call("jova_gemm_f32_img", A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, float(alpha), bias, residual, ldr, int(relu),
             splitk, nb1, nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], P(img), stream())
        return
    call("jova_gemm_f32", A, sam, sak, B, sbk, sbn, C, ldc, M, N, K, float(alpha), bias, residual, ldr, int(relu),
         splitk, nb1, nb2, sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], stream())


def _b_image(A, sam, sak, B, sbk, sbn, M, N, K, splitk=1):
    """The bf16-pieces image of B for a product of a GFLOP or more that jova_gemm_f32_img would run with it (None
    otherwise): one small launch, 6 N K bytes from torch's caching allocator."""
    if M * N * K < (1 << 30) or sak != 1:
        return None
    need = _lib.lib().jova_gemm_b_image_bytes(A, sam, sak, B, sbk, sbn, M, N, K, 1, 1, splitk)
    if not need:
        return None
    img = torch.empty(need, dtype=torch.uint8, device=torch.device("cuda", torch.cuda.current_device()))
    call("jova_gemm_b_image", B, sbk, sbn, N, K, P(img), stream())
    return img


def linear_fwd(x2d, w, b, out2d, relu=False, residual=None, tag=None):
    """out[M,N] = x[M,K] @ w[N,K]^T + b (+ residual) (ReLU)."""
    M, K = x2d.shape
    N = w.shape[0]
    gemm(_addr(x2d), K, 1, _addr(w), 1, K, _addr(out2d), N, M, N, K, bias=P(b) if b is not None else None,
         residual=_addr(residual) if residual is not None else None, ldr=N, relu=relu, tag=tag)


def linear_bwd_data(dy2d, w, dx2d, residual=None, relu_mask=None):
    """dx[M,K] = dy[M,N] @ w[N,K] (+ residual), or zeroed where relu_mask[M,K] <= 0 (fused ReLU backward)."""
    M, N = dy2d.shape
    K = w.shape[1]
    if relu_mask is not None:
        if residual is not None:
            raise ValueError("linear_bwd_data: residual and relu_mask exclude each other")
        gemm(_addr(dy2d), N, 1, _addr(w), K, 1, _addr(dx2d), K, M, K, N, residual=_addr(relu_mask), ldr=K, relu=2)
        return
    gemm(_addr(dy2d), N, 1, _addr(w), K, 1, _addr(dx2d), K, M, K, N,
         residual=_addr(residual) if residual is not None else None, ldr=K)


def _splitk_for(out_rows, out_cols, k):
    """Split of a weight-gradient product's contraction (the rows) over workgroups.  (Round 3 swept the three constants on the
    thin gradients of the Davis step under a kernel trace -- tools/bench_small_gemm.py: 2048 / 128 / 128 and 4096 / 128 /
    256 move single launches by -12 .. +9 us and the sum by 3 %: left as they were.)"""
    blocks = ((out_rows + 127) // 128) * ((out_cols + 127) // 128)
    sk = max(1, min(512 // max(blocks, 1), (k + 255) // 256))
    return min(sk, 64)


def linear_bwd_weight(dy2d, x2d, dw, db):
    """dw[N,K] += dy[M,N]^T @ x[M,K];  db[N] += column sums of dy."""
    M, N = dy2d.shape
    K = x2d.shape[1]
    if dw is not None:
        # (split-K adds atomically into dw; a single split accumulates through the residual; the bias gradient rides on
        # the product's reads of dy when the aligned form runs)
        call("jova_linear_bwd_weight", P(dy2d), N, P(x2d), K, M, N, K, P(dw), P(db), _splitk_for(N, K, M), stream())
    elif db is not None:
        call("jova_colsum_add", P(dy2d), M, N, N, P(db), stream())


class LinearFunction(torch.autograd.Function):
    """nn.Linear (+ optional fused ReLU) over the last dimension."""

    @staticmethod
    def forward(ctx, x, weight, bias, relu):
        x = _c(x)
        k = x.shape[-1]
        x2d = x.reshape(-1, k)
        out = _empty(x.shape[:-1] + (weight.shape[0],), x)
        linear_fwd(x2d, weight, bias, out.view(-1, weight.shape[0]), relu=relu)
        ctx.relu = relu
        ctx.params = (weight, bias)
        ctx.save_for_backward(x2d, out if relu else None)
        ctx.xshape = x.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        x2d, out = ctx.saved_tensors
        weight, bias = ctx.params
        dy = _c(dy)
        n = weight.shape[0]
        dy2d = dy.reshape(-1, n)
        if ctx.relu:
            dz = torch.empty_like(dy2d)
            call("jova_relu_bwd", P(dy2d), P(out.view(-1, n)), dy2d.numel(), P(dz), stream())
            dy2d = dz
        sink = _GradSink((weight, bias))
        linear_bwd_weight(dy2d, x2d, sink.targets[0], sink.targets[1])
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty(x2d.shape, dy)
            linear_bwd_data(dy2d, weight, dx)
            dx = dx.view(ctx.xshape)
        return dx, sink.returns[0], sink.returns[1], None


class BatchNormFunction(torch.autograd.Function):
    """nn.BatchNorm1d in training mode over [rows, C] (+ optional fused ReLU); running stats updated in-kernel."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, num_batches, eps, momentum, relu, stats,
                input_relu=False):
        x = _c(x)
        n, c = x.shape
        if n < 2:
            raise ValueError("Expected more than 1 value per channel when training, got input size %s" % (tuple(x.shape),))
        if stats is None:
            stats = _empty((2, c), x, torch.float64)
            call("jova_colstats", P(x), n, c, P(stats), stream())
        y = torch.empty_like(x)
        mean_rstd = _empty((2, c), x)
        call("jova_batchnorm_fwd", P(x), n, c, P(stats), P(gamma), P(beta), float(eps), float(momentum),
             P(running_mean), P(running_var), P(num_batches), int(relu), P(y), P(mean_rstd), stream())
        ctx.relu = relu
        ctx.input_relu = bool(input_relu)     # x is the sole-consumer output of a ReLU: its backward is fused below
        ctx.params = (gamma, beta)
        ctx.save_for_backward(x, y if relu else None, mean_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, mean_rstd = ctx.saved_tensors
        gamma, beta = ctx.params
        dy = _c(dy)
        n, c = x.shape
        sink = _GradSink((gamma, beta))
        dg = sink.targets[0] if sink.targets[0] is not None else _zeros((c,), x)
        db = sink.targets[1] if sink.targets[1] is not None else _zeros((c,), x)
        dx = torch.empty_like(x)
        ws = _empty((2, c), x, torch.float64)
        call("jova_batchnorm_bwd", P(x), P(dy), P(y) if ctx.relu else None, P(mean_rstd), P(gamma), n, c,
             int(ctx.input_relu), P(dx), P(dg), P(db), P(ws), stream())
        return dx, sink.returns[0], sink.returns[1], None, None, None, None, None, None, None, None


def batch_norm_eval(x, gamma, beta, running_mean, running_var, eps, relu):
    """Inference-mode BatchNorm1d through the same apply kernel (mean / rstd from the running statistics)."""
    x = _c(x)
    n, c = x.shape
    mean_rstd = torch.stack([running_mean, torch.rsqrt(running_var + eps)]).contiguous()
    # a zero-variance-free way to reuse jova_batchnorm_fwd would recompute stats; the apply kernel is reached by
    # feeding statistics that reproduce mean_rstd exactly: sum = n*mean, sumsq = n*(var + mean^2)
    stats = torch.stack([running_mean.double() * n, (running_var.double() + running_mean.double() ** 2) * n])
    y = torch.empty_like(x)
    call("jova_batchnorm_fwd", P(x), n, c, P(stats.contiguous()), P(gamma), P(beta), float(eps), 0.0, None, None, None,
         int(relu), P(y), P(mean_rstd), stream())
    return y


class ReLUFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        y = torch.empty_like(x)
        call("jova_relu_fwd", P(x), x.numel(), P(y), stream())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        call("jova_relu_bwd", P(dy), P(y), dy.numel(), P(dx), stream())
        return dx


class _DropoutState(object):
    """Philox (seed, offset) bookkeeping: one offset per dropout call.  The key is torch's global seed mixed with a
    per-rank stream id (set_dropout_stream; the Trainer passes its rank), so data-parallel ranks that share
    ``torch.manual_seed`` (identical initial weights) still draw independent masks."""
    offset = 0
    stream = 0

    @classmethod
    def next(cls):
        cls.offset += 1
        seed = (torch.initial_seed() ^ (cls.stream * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
        return seed, cls.offset


def set_dropout_stream(stream_id):
    """Selects the dropout mask stream of this process (rank of a data-parallel job); restarts the offset counter."""
    _DropoutState.stream = int(stream_id)
    _DropoutState.offset = 0


def dropout_apply(x2d_addr, rows, n, ld, p, seed, offset, out_addr):
    call("jova_dropout", x2d_addr, rows, n, ld, float(p), seed, offset, out_addr, stream())


class DropoutFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        y = torch.empty_like(x)
        ctx.key = _DropoutState.next()
        ctx.p = p
        dropout_apply(P(x), 1, x.numel(), x.numel(), p, ctx.key[0], ctx.key[1], P(y))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        dx = torch.empty_like(dy)
        dropout_apply(P(dy), 1, dy.numel(), dy.numel(), ctx.p, ctx.key[0], ctx.key[1], P(dx))
        return dx, None


# ------------------------------------------------------------------------------------------------------ protein
class Prot2VecFunction(torch.autograd.Function):
    """Embedding gather + reshape + activation + permute of Prot2Vec.forward (jova/nn/models.py:704-715)."""

    @staticmethod
    def forward(ctx, emb, table, prot_off, prot_len, p_max, window, relu):
        bsz = prot_off.shape[0]
        vocab, e = emb.shape[0] - 1, emb.shape[1]
        out = _empty((p_max, bsz, window * e), emb)
        call("jova_prot2vec_fwd", P(table), P(prot_off), P(prot_len), P(emb), bsz, p_max, window, e, vocab, int(relu),
             P(out), None, stream())
        ctx.args = (table, prot_off, prot_len, p_max, window, relu, bsz, vocab, e)
        ctx.params = (emb,)
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        (out,) = ctx.saved_tensors
        table, prot_off, prot_len, p_max, window, relu, bsz, vocab, e = ctx.args
        dout = _c(dout)
        sink = _GradSink(ctx.params)
        if sink.targets[0] is not None:
            call("jova_prot2vec_bwd", P(table), P(prot_off), P(prot_len), P(dout), P(out), bsz, p_max, window, e,
                 vocab, int(relu), P(sink.targets[0]), stream())
        return sink.returns[0], None, None, None, None, None, None


def copy2d(src, src_rows, src_cols, ld_src, dst, dst_rows, dst_cols, ld_dst, accumulate=False):
    call("jova_copy2d", P(src), src_rows, src_cols, ld_src, P(dst), dst_rows, dst_cols, ld_dst, int(accumulate), stream())


class ProteinCnnFunction(torch.autograd.Function):
    """ProteinCNN / ProteinCNN2D's stack  x -> relu(W_k x + b_k)  (jova/nn/models.py:738-739, :768-771) as one autograd
    node whose square layers run on the ALIGNED fp32 GEMM: the window * dim = 242 wide activations live in buffers padded
    to 256 columns (zeros), the [242, 242] weights are padded per call into [256, 256] images (zero rows / columns keep
    the padding at zero through every layer, forward and backward), so all six 23 k x 242 x 242 products of a Davis step
    take the LDS-DMA kernel (110 instead of 38 TFLOP/s) at 12 % more arithmetic.  The last layer (242 -> dim) and its
    gradients read the padded buffer with its row stride.  Used when rows % 128 == 0 (B = 128); otherwise the layers run
    one by one on the generic form (LinearFunction)."""

    @staticmethod
    def usable(x, weights):
        d = x.shape[-1]
        m = x.numel() // max(d, 1)
        dp = (d + 127) // 128 * 128
        return (len(weights) >= 2 and m > 0 and m % 128 == 0 and d % 128 != 0 and dp * dp <= 1.3 * d * d and
                all(tuple(w.shape) == (d, d) for w in weights[:-1]) and weights[-1].shape[1] == d)

    @staticmethod
    def forward(ctx, x, *params):
        ws, bs = params[0::2], params[1::2]
        x = _c(x)
        d = x.shape[-1]
        m = x.numel() // d
        dp = (d + 127) // 128 * 128
        n_out = ws[-1].shape[0]
        h = _empty((m, dp), x)
        copy2d(x, m, d, d, h, m, dp, dp)
        hs, wps = [h], []
        for w, b in zip(ws[:-1], bs[:-1]):
            wp, bp = _empty((dp, dp), x), _empty((dp,), x)
            copy2d(w, d, d, d, wp, dp, dp, dp)
            copy2d(b, 1, d, d, bp, 1, dp, dp)
            nxt = _empty((m, dp), x)
            linear_fwd(hs[-1], wp, bp, nxt, relu=True)
            hs.append(nxt)
            wps.append(wp)
        out = _empty(x.shape[:-1] + (n_out,), x)
        gemm(_addr(hs[-1]), dp, 1, _addr(ws[-1]), 1, d, _addr(out), n_out, m, n_out, d, bias=P(bs[-1]), relu=True)
        ctx.params, ctx.hs, ctx.wps, ctx.out = params, hs, wps, out
        ctx.dims = (m, d, dp, n_out, x.shape)
        return out

    @staticmethod
    def backward(ctx, dout):
        params, hs, wps, out = ctx.params, ctx.hs, ctx.wps, ctx.out
        m, d, dp, n_out, xshape = ctx.dims
        ws = params[0::2]
        sink = _GradSink(params)
        dout = _c(dout).reshape(m, n_out)
        dz = torch.empty_like(dout)
        call("jova_relu_bwd", P(dout), P(out.view(m, n_out)), dout.numel(), P(dz), stream())
        # last layer: dW [n_out, d] += dz^T h[:, :d];  db += colsum(dz);  dh = (dz W) masked by the ReLU of h, into padded rows
        last = len(ws) - 1
        sk = _splitk_for(n_out, d, m)
        if sk > 1:
            gemm(_addr(dz), 1, n_out, _addr(hs[-1]), dp, 1, _addr(sink.targets[2 * last]), d, n_out, d, m, splitk=sk)
        else:
            gemm(_addr(dz), 1, n_out, _addr(hs[-1]), dp, 1, _addr(sink.targets[2 * last]), d, n_out, d, m,
                 residual=_addr(sink.targets[2 * last]), ldr=d)
        call("jova_colsum_add", P(dz), m, n_out, n_out, P(sink.targets[2 * last + 1]), stream())
        need_dx = ctx.needs_input_grad[0]
        dh = None
        if last > 0 or need_dx:
            dh = _zeros((m, dp), dout)
            if last > 0:
                gemm(_addr(dz), n_out, 1, _addr(ws[-1]), d, 1, _addr(dh), dp, m, d, n_out, residual=_addr(hs[-1]), ldr=dp, relu=2)
            else:
                gemm(_addr(dz), n_out, 1, _addr(ws[-1]), d, 1, _addr(dh), dp, m, d, n_out)
        for k in range(last - 1, -1, -1):
            # h[k+1] = relu(h[k] Wp_k^T + bp_k); dh is already masked with h[k+1] > 0
            dwp, dbp = _zeros((dp, dp), dout), _zeros((dp,), dout)
            call("jova_linear_bwd_weight", P(dh), dp, P(hs[k]), dp, m, dp, dp, P(dwp), P(dbp), _splitk_for(dp, dp, m), stream())
            copy2d(dwp, d, d, dp, sink.targets[2 * k], d, d, d, accumulate=True)
            copy2d(dbp, 1, d, dp, sink.targets[2 * k + 1], 1, d, d, accumulate=True)
            if k > 0 or need_dx:
                nxt = _empty((m, dp), dout)
                if k > 0:
                    linear_bwd_data(dh, wps[k], nxt, relu_mask=hs[k])
                else:
                    linear_bwd_data(dh, wps[k], nxt)
                dh = nxt
        dx = None
        if need_dx:
            dx = _empty((m, d), dout)
            copy2d(dh, m, d, dp, dx, m, d, d)
            dx = dx.view(xshape)
        ctx.hs = ctx.wps = ctx.out = None
        return (dx,) + tuple(sink.returns)


# ------------------------------------------------------------------------------------------------------ joint attention
class JointAttentionFunction(torch.autograd.Function):
    """JointAttention.forward (jova/nn/models.py:939-973) up to and including the view pooling, as one autograd node.

    params per layer i (order): in_proj_weight, in_proj_bias, out_proj.weight, out_proj.bias,
                                linear1.weight, linear1.bias, linear2.weight, linear2.bias, normalize.weight, normalize.bias
    head of the list: lin_prjs[v].weight, lin_prjs[v].bias for every view, then attn_norm.weight, attn_norm.bias.
    """

    @staticmethod
    def forward(ctx, n_views, num_heads, num_layers, dprob, training, attn_hook, *args):
        views = [_c(v) for v in args[:n_views]]
        params = args[n_views:]
        prj = params[:2 * n_views]
        attn_g, attn_b = params[2 * n_views], params[2 * n_views + 1]
        layers = [params[2 * n_views + 2 + 10 * i: 2 * n_views + 2 + 10 * (i + 1)] for i in range(num_layers)]
        bsz = views[0].shape[1]
        L = prj[0].shape[0]
        segs = [v.shape[0] for v in views]
        S = sum(segs)
        T = S * bsz
        H = num_heads
        hd = L // H
        Sp = (S + 3) // 4 * 4
        ref = views[0]
        seg_off = [0]
        for s in segs:
            seg_off.append(seg_off[-1] + s)
        p_drop = float(dprob) if training else 0.0

        x = _empty((T, L), ref)
        for v in range(n_views):
            xv = views[v].reshape(-1, views[v].shape[2])
            linear_fwd(xv, prj[2 * v], prj[2 * v + 1], x[seg_off[v] * bsz: seg_off[v + 1] * bsz])
        saved = []
        for lp in layers:
            w_in, b_in, w_o, b_o, w1, b1, w2, b2, g2, be2 = lp
            qkv = _empty((T, 3 * L), ref)
            linear_fwd(x, w_in, b_in, qkv)
            probs = _empty((bsz * H, S, Sp), ref)
            # scores[b,h][s][s'] = (q . k) / sqrt(hd)
            gemm(_addr(qkv), bsz * 3 * L, 1, _addr(qkv, L), 1, bsz * 3 * L, _addr(probs), Sp, S, S, hd,
                 alpha=1.0 / math.sqrt(hd), nb1=bsz, nb2=H, sA=(3 * L, hd), sB=(3 * L, hd), sC=(H * S * Sp, S * Sp))
            call("jova_softmax_fwd", P(probs), bsz * H * S, S, Sp, P(probs), stream())
            key = None
            pd = probs
            if p_drop > 0:
                key = _DropoutState.next()
                pd = torch.empty_like(probs)
                dropout_apply(P(probs), bsz * H * S, S, Sp, p_drop, key[0], key[1], P(pd))
            o = _empty((T, L), ref)
            # o[s][b][h*hd + d] = sum_s' pd[b,h][s][s'] v[s'][b][h*hd + d]
            gemm(_addr(pd), Sp, 1, _addr(qkv, 2 * L), bsz * 3 * L, 1, _addr(o), bsz * L, S, hd, S, nb1=bsz, nb2=H,
                 sA=(H * S * Sp, S * Sp), sB=(3 * L, hd), sC=(L, hd))
            xp = _empty((T, L), ref)
            linear_fwd(o, w_o, b_o, xp)
            if attn_hook is not None:
                # models.py:952-956: hook(layer, x, x', head-averaged weights [batch, S, S], segments per view); the
                # weights are what nn.MultiheadAttention returns (after its dropout, averaged over the heads)
                wts = pd.view(bsz, H, S, Sp)[..., :S].mean(dim=1)
                attn_hook(len(saved), x.view(S, bsz, L), xp.view(S, bsz, L), wts, list(segs))
            x1 = _empty((T, L), ref)
            mr1 = _empty((T, 2), ref)
            call("jova_layernorm_fwd", P(x), P(xp), P(attn_g), P(attn_b), T, L, 1e-5, P(x1), P(mr1), stream())
            hid = _empty((T, w1.shape[0]), ref)
            linear_fwd(x1, w1, b1, hid, relu=True, tag="ffn1_fwd")
            f = _empty((T, L), ref)
            linear_fwd(hid, w2, b2, f)
            x2 = _empty((T, L), ref)
            mr2 = _empty((T, 2), ref)
            call("jova_layernorm_fwd", P(f), P(x1), P(g2), P(be2), T, L, 1e-5, P(x2), P(mr2), stream())
            saved.append((x, qkv, probs, o, xp, x1, mr1, hid, f, mr2, pd if p_drop > 0 else None))
            x = x2
        seg_off_dev = _seg_off_device(seg_off, ref.device)
        out = _empty((bsz, n_views * L), ref)
        call("jova_viewpool_fwd", P(x), P(seg_off_dev), n_views, bsz, L, P(out), stream())
        ctx.cfg = (n_views, H, num_layers, p_drop, bsz, L, segs, S, Sp, T, hd, seg_off, seg_off_dev)
        ctx.views, ctx.params, ctx.saved = views, params, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        n_views, H, num_layers, p_drop, bsz, L, segs, S, Sp, T, hd, seg_off, seg_off_dev = ctx.cfg
        views, params, saved = ctx.views, ctx.params, ctx.saved
        ref = views[0]
        sink = _GradSink(params)
        tg = sink.targets
        prj = params[:2 * n_views]
        attn_g = params[2 * n_views]
        dout = _c(dout)
        dx = _empty((T, L), ref)
        call("jova_viewpool_bwd", P(dout), P(seg_off_dev), n_views, bsz, L, S, P(dx), stream())
        scale = 1.0 / math.sqrt(hd)
        for i in reversed(range(num_layers)):
            base = 2 * n_views + 2 + 10 * i
            w_in, b_in, w_o, b_o, w1, b1, w2, b2, g2, be2 = params[base: base + 10]
            x, qkv, probs, o, xp, x1, mr1, hid, f, mr2, pd_saved = saved[i]
            inner = w1.shape[0]
            # LN2: z2 = f + x1
            dz2 = _empty((T, L), ref)
            call("jova_layernorm_bwd", P(f), P(x1), P(dx), P(g2), P(mr2), T, L, P(dz2), P(tg[base + 8]),
                 P(tg[base + 9]), stream())
            # FFN: f = hid @ w2^T + b2 ; hid = relu(x1 @ w1^T + b1)
            linear_bwd_weight(dz2, hid, tg[base + 6], tg[base + 7])
            dhid = _empty((T, inner), ref)
            linear_bwd_data(dz2, w2, dhid, relu_mask=hid)          # ReLU backward fused into the epilogue
            linear_bwd_weight(dhid, x1, tg[base + 4], tg[base + 5])
            dx1 = _empty((T, L), ref)
            linear_bwd_data(dhid, w1, dx1, residual=dz2)          # + residual branch of LN2
            del dhid
            # LN1: z1 = x + xp (shared attn_norm)
            dz1 = _empty((T, L), ref)
            call("jova_layernorm_bwd", P(x), P(xp), P(dx1), P(attn_g), P(mr1), T, L, P(dz1), P(tg[2 * n_views]),
                 P(tg[2 * n_views + 1]), stream())
            # out_proj: xp = o @ w_o^T + b_o
            linear_bwd_weight(dz1, o, tg[base + 2], tg[base + 3])
            do = _empty((T, L), ref)
            linear_bwd_data(dz1, w_o, do)
            # attention core
            # (the dropped probabilities are kept from the forward: 80 MB at Davis size, two passes less than
            # regenerating the mask twice)
            pd = pd_saved if p_drop > 0 else probs
            dqkv = _empty((T, 3 * L), ref)
            # dv[s'][d] = sum_s pd[s][s'] do[s][d]
            gemm(_addr(pd), 1, Sp, _addr(do), bsz * L, 1, _addr(dqkv, 2 * L), bsz * 3 * L, S, hd, S, nb1=bsz, nb2=H,
                 sA=(H * S * Sp, S * Sp), sB=(L, hd), sC=(3 * L, hd))
            # dp[s][s'] = sum_d do[s][d] v[s'][d]
            dp = torch.empty_like(probs)
            gemm(_addr(do), bsz * L, 1, _addr(qkv, 2 * L), 1, bsz * 3 * L, _addr(dp), Sp, S, S, hd, nb1=bsz, nb2=H,
                 sA=(L, hd), sB=(3 * L, hd), sC=(H * S * Sp, S * Sp))
            # dropout backward (mask from pd, scale 1 / (1 - p)) rides in the softmax backward
            call("jova_softmax_bwd", P(probs), P(dp), P(pd) if p_drop > 0 else None, 1.0 / (1.0 - p_drop), bsz * H * S, S,
                 Sp, P(dp), stream())
            # dq[s][d] = scale * sum_s' ds[s][s'] k[s'][d] ;  dk[s'][d] = scale * sum_s ds[s][s'] q[s][d]
            gemm(_addr(dp), Sp, 1, _addr(qkv, L), bsz * 3 * L, 1, _addr(dqkv), bsz * 3 * L, S, hd, S, alpha=scale,
                 nb1=bsz, nb2=H, sA=(H * S * Sp, S * Sp), sB=(3 * L, hd), sC=(3 * L, hd))
            gemm(_addr(dp), 1, Sp, _addr(qkv), bsz * 3 * L, 1, _addr(dqkv, L), bsz * 3 * L, S, hd, S, alpha=scale,
                 nb1=bsz, nb2=H, sA=(H * S * Sp, S * Sp), sB=(3 * L, hd), sC=(3 * L, hd))
            del dp, pd
            linear_bwd_weight(dqkv, x, tg[base], tg[base + 1])
            dx = _empty((T, L), ref)
            linear_bwd_data(dqkv, w_in, dx, residual=dz1)         # + residual branch of LN1
        # view projections
        grads_in = []
        for v in range(n_views):
            dxv = dx[seg_off[v] * bsz: seg_off[v + 1] * bsz]
            xv = views[v].reshape(-1, views[v].shape[2])
            linear_bwd_weight(dxv, xv, tg[2 * v], tg[2 * v + 1])
            if ctx.needs_input_grad[6 + v]:
                gv = _empty(xv.shape, ref)
                linear_bwd_data(dxv, prj[2 * v], gv)
                grads_in.append(gv.view(views[v].shape))
            else:
                grads_in.append(None)
        ctx.saved = None
        return (None, None, None, None, None, None) + tuple(grads_in) + tuple(sink.returns)


class EmbeddingFunction(torch.autograd.Function):
    """nn.Embedding lookup of int64 ids (GraphNeuralNet.embed_fingerprint, jova/nn/models.py:362, :376)."""
    bad_flag = None          # device int: an id outside [0, vocab) was seen (nn.Embedding raises there); see raise_if_bad

    @staticmethod
    def forward(ctx, weight, idx):
        _lib.require_gpu(weight, idx)
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        n, d = idx.numel(), weight.shape[1]
        if EmbeddingFunction.bad_flag is None or EmbeddingFunction.bad_flag.device != weight.device:
            EmbeddingFunction.bad_flag = torch.zeros(1, dtype=torch.int32, device=weight.device)
        out = _empty((n, d), weight)
        call("jova_embedding_fwd", P(weight), P(idx), n, d, weight.shape[0], P(out), P(EmbeddingFunction.bad_flag), stream())
        ctx.idx, ctx.weight = idx, weight
        return out

    @staticmethod
    def backward(ctx, dout):
        sink = _GradSink((ctx.weight,))
        dout = _c(dout)
        call("jova_embedding_bwd", P(dout), P(ctx.idx), ctx.idx.numel(), ctx.weight.shape[1], ctx.weight.shape[0],
             P(sink.targets[0]), stream())
        return sink.returns[0], None

    @staticmethod
    def raise_if_bad():
        """One device->host read, off the hot loop: IndexError if any lookup so far had an id outside the table."""
        f = EmbeddingFunction.bad_flag
        if f is not None and int(f.item()):
            f.zero_()
            raise IndexError("index out of range in self (fingerprint id outside the embedding table)")


class BlockAdjacency(object):
    """The batch's block-diagonal adjacency of the gnn view as its non-zeros (CSR of A and of A^T, device int32 / fp32)
    instead of the reference's dense [N, N] matrix (jova/data/data.py:309-335: 67 MB at N = 4.1 k, 99.9 % zeros).
    ``to_dense()`` gives the reference's tensor.  Made by jova_amd.data.data.process_gnn_view_data."""

    def __init__(self, n, row_ptr, col, val, row_ptr_t, col_t, val_t):
        self.n, self.shape = int(n), (int(n), int(n))
        self.row_ptr, self.col, self.val = row_ptr, col, val
        self.row_ptr_t, self.col_t, self.val_t = row_ptr_t, col_t, val_t
        self.device, self.is_cuda = row_ptr.device, row_ptr.is_cuda

    def to_dense(self):
        rp = self.row_ptr.cpu().numpy()
        rows = np.repeat(np.arange(self.n), np.diff(rp))
        a = torch.zeros(self.shape, dtype=_F32)
        a[torch.from_numpy(rows), self.col.cpu().long()] = self.val.cpu()
        return a.to(self.device)


class GnnUpdateFunction(torch.autograd.Function):
    """x + (A @ hs) / (M - 1)  ('mean')  or  x + A @ hs  ('sum'): GraphNeuralNet.update_fingerprints after its Linear + ReLU
    (jova/nn/models.py:364-372).  A (no gradient) is either the dense block-diagonal matrix the reference's collator hands
    over ([N, N] fp32: one fp32 MFMA GEMM forward, one with A^T backward) or a BlockAdjacency (this package's collator:
    the non-zeros only, jova_spmm_csr forward and backward)."""

    @staticmethod
    def forward(ctx, x, hs, A, M):
        x, hs = _c(x), _c(hs)
        n, d = x.shape
        if tuple(A.shape) != (n, n):
            raise RuntimeError("adjacency is %s for %d nodes" % (tuple(A.shape), n))
        m = None if M is None else _c(M).reshape(-1)
        tmp = _empty((n, d), x)
        if isinstance(A, BlockAdjacency):
            call("jova_spmm_csr", P(A.row_ptr), P(A.col), P(A.val), P(hs), n, d, n, P(tmp), stream())
        else:
            A = _c(A)
            gemm(_addr(A), n, 1, _addr(hs), d, 1, _addr(tmp), d, n, d, n)
        out = _empty((n, d), x)
        call("jova_rowscale_add", P(x), P(tmp), P(m), n, d, P(out), stream())
        ctx.A, ctx.m = A, m
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        n, d = dout.shape
        g = dout
        if ctx.m is not None:
            g = torch.empty_like(dout)
            call("jova_rowscale_add", None, P(dout), P(ctx.m), n, d, P(g), stream())
        dhs = _empty((n, d), dout)
        A = ctx.A
        if isinstance(A, BlockAdjacency):
            call("jova_spmm_csr", P(A.row_ptr_t), P(A.col_t), P(A.val_t), P(g), n, d, n, P(dhs), stream())     # A^T g
        else:
            gemm(_addr(A), 1, n, _addr(g), d, 1, _addr(dhs), d, n, d, n)          # A^T g
        return dout, dhs, None, None


class LstmFunction(torch.autograd.Function):
    """Single-layer one-directional nn.LSTM with zero initial state followed by ReLU (ProteinRNN.forward,
    jova/nn/models.py:672-682): x [T, B, in] -> relu(h) [T, B, H].  The input part of the gates is one GEMM over all steps,
    the recurrence one persistent launch (jova_lstm_fwd); backward: jova_lstm_bwd for the gate gradients, then GEMMs."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, relu):
        x = _c(x)
        T, B, K = x.shape
        G, H = w_hh.shape
        bias = b_ih + b_hh if b_ih is not None else None           # (4H floats; parameters are read, not written)
        gx = _empty((T * B, G), x)
        linear_fwd(x.view(T * B, K), w_ih, bias, gx)
        h, gates, c = _empty((T, B, H), x), _empty((T, B, G), x), _empty((T, B, H), x)
        call("jova_lstm_fwd", P(gx), P(w_hh), T, B, H, P(h), P(gates), P(c), stream())
        out = h
        if relu:
            out = torch.empty_like(h)
            call("jova_relu_fwd", P(h), h.numel(), P(out), stream())
        ctx.params = (w_ih, w_hh, b_ih, b_hh)
        ctx.saved = (x, h, gates, c, out if relu else None)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, h, gates, c, out = ctx.saved
        w_ih, w_hh, b_ih, b_hh = ctx.params
        T, B, K = x.shape
        G, H = w_hh.shape
        dout = _c(dout)
        dh = dout
        if out is not None:
            dh = torch.empty_like(dout)
            call("jova_relu_bwd", P(dout), P(out), dout.numel(), P(dh), stream())
        dgx = _empty((T, B, G), x)
        call("jova_lstm_bwd", P(dh), P(w_hh), P(gates), P(c), T, B, H, P(dgx), stream())
        sink = _GradSink(ctx.params)
        dg2 = dgx.view(T * B, G)
        linear_bwd_weight(dg2, x.view(T * B, K), sink.targets[0], sink.targets[2])
        if sink.targets[3] is not None:
            call("jova_colsum_add", P(dg2), T * B, G, G, P(sink.targets[3]), stream())
        if T > 1:                                                   # dW_hh = sum_t dg_t^T h_{t-1}  (h_{-1} = 0)
            linear_bwd_weight(dgx[1:].reshape((T - 1) * B, G), h[:T - 1].reshape((T - 1) * B, H), sink.targets[1], None)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _empty((T * B, K), x)
            linear_bwd_data(dg2, w_ih, dx)
            dx = dx.view(T, B, K)
        ctx.saved = None
        return dx, sink.returns[0], sink.returns[1], sink.returns[2], sink.returns[3], None


class RowGatherFunction(torch.autograd.Function):
    """src[idx] over the first dimension (atom_features[atom_to_pair], jova/nn/layers.py:141-142); backward = scatter-add."""

    @staticmethod
    def forward(ctx, src, idx):
        src = _c(src)
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        m, d = src.shape
        out = _empty((idx.numel(), d), src)
        call("jova_gather_rows", P(src), P(idx), idx.numel(), d, m, P(out), stream())
        ctx.idx, ctx.m = idx, m
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        dsrc = _zeros((ctx.m, dout.shape[1]), dout)
        call("jova_scatter_add_rows", P(dout), P(ctx.idx), ctx.idx.numel(), dout.shape[1], ctx.m, P(dsrc), stream())
        return dsrc, None


class RowScatterAddFunction(torch.autograd.Function):
    """torch_scatter.scatter_add(src, index, dim=0, fill_value=0) (the call of jova/nn/layers.py:135, :191): rows of src
    summed into out[index]; out has max(index) + 1 rows unless ``m`` says otherwise.  Backward = gather."""

    @staticmethod
    def forward(ctx, src, idx, m):
        src = _c(src)
        idx = idx.reshape(-1).to(torch.int64).contiguous()
        out = _zeros((m, src.shape[1]), src)
        call("jova_scatter_add_rows", P(src), P(idx), idx.numel(), src.shape[1], m, P(out), stream())
        ctx.idx, ctx.m = idx, m
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = _c(dout)
        dsrc = _empty((ctx.idx.numel(), dout.shape[1]), dout)
        call("jova_gather_rows", P(dout), P(ctx.idx), ctx.idx.numel(), dout.shape[1], ctx.m, P(dsrc), stream())
        return dsrc, None, None


class GaussHistFunction(torch.autograd.Function):
    """WeaveGather.gaussian_histogram (jova/nn/layers.py:198-207): [n, c] -> [n, 11 c]."""

    @staticmethod
    def forward(ctx, x, eps):
        x = _c(x)
        n, c = x.shape
        out = _empty((n, 11 * c), x)
        call("jova_gauss_hist_fwd", P(x), x.numel(), float(eps), P(out), stream())
        ctx.eps = float(eps)
        ctx.save_for_backward(x)
        return out

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        dout = _c(dout)
        dx = torch.empty_like(x)
        call("jova_gauss_hist_bwd", P(x), P(dout), x.numel(), ctx.eps, P(dx), stream())
        return dx, None


class MSELossFunction(torch.autograd.Function):
    """nn.MSELoss()(outputs * weights, target) (proj/dti/train_joint_attn.py:464-465)."""

    @staticmethod
    def forward(ctx, outputs, weights, target):
        if not (outputs.dtype == weights.dtype == target.dtype == _F32):
            raise TypeError("MSELossFunction takes float32 outputs, weights and target, got %s/%s/%s"
                            % (outputs.dtype, weights.dtype, target.dtype))
        if not (outputs.shape == weights.shape == target.shape):
            raise ValueError("MSELossFunction: outputs, weights and target must have one shape")
        o, w, y = _c(outputs), _c(weights), _c(target)
        loss = _empty((), o)
        dout = torch.empty_like(o)
        call("jova_mse_loss", P(o), P(w), P(y), o.numel(), P(loss), P(dout), stream())
        ctx.save_for_backward(dout)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        (dout,) = ctx.saved_tensors
        g = torch.empty_like(dout)
        call("jova_scale_dev", P(dout), P(_c(dloss.reshape(1).to(dout.dtype))), dout.numel(), P(g), stream())
        return g, None, None
