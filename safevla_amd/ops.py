"""Stream-aware Python bindings of the C-ABI kernels (include/svla.h).  torch is plumbing only: device memory
and the current HIP stream.  Every function launches on ``torch.cuda.current_stream()`` and raises on failure."""
import ctypes
import struct
from typing import Optional, Tuple

import torch

from ._lib import lib

BF16 = torch.bfloat16
F32 = torch.float32

MASK_NONE, MASK_BLOCK_CAUSAL = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


class LaunchPlan:
    """A recorded sequence of C-ABI calls (bound function + fully converted arguments, stream handle included) that can be re-issued from
    one tight loop: no tensor allocation, no wrapper code, no argument conversion.  Everything a recorded call touched -- tensors, dropout
    descriptors -- is kept alive by the plan, so the recorded device pointers stay valid; step-dependent state must live in device memory
    (svla_dropout.seed_dev, the KV-cache slot of svla_kv_append_bf16).  Used for the single-step acting forward, which is ~100 small
    dependent kernels per tower: issuing them through the Python wrappers costs ~14 us per launch, replaying them ~6.5 us
    (tools/replay_probe.py).  HIP graphs, the textbook tool, replay slower than eager issue on this stack."""

    def __init__(self):
        self.calls, self.keep = [], []

    def __enter__(self):
        global _REC
        assert _REC is None and lib().recorder is None, "launch recording is not re-entrant"
        _REC = self
        lib().recorder = self.calls
        return self

    def __exit__(self, *exc):
        global _REC
        _REC = None
        lib().recorder = None

    def compile(self):
        """Flatten the recorded calls for svla_replay_calls (include/svla.h): one FFI crossing per replay instead of one per launch.
        Arguments become 64-bit words -- pointers / integers by value, float / double by bit pattern, per the header's declaration."""
        L = lib()
        ids, offs, words = [], [], []
        for fn, a in self.calls:
            name = fn.__name__
            decl = L.decls[name]
            assert len(decl) == len(a), name
            ids.append(L.fn_ids[name])
            offs.append(len(words))
            for (_, ct), v in zip(decl, a):
                if ct is ctypes.c_float:
                    words.append(struct.unpack("<I", struct.pack("<f", float(v)))[0])
                elif ct is ctypes.c_double:
                    words.append(struct.unpack("<Q", struct.pack("<d", float(v)))[0])
                elif v is None:
                    words.append(0)
                elif isinstance(v, ctypes.c_void_p):
                    words.append(int(v.value or 0))
                else:
                    words.append(int(v) & 0xFFFFFFFFFFFFFFFF)
        self._n = len(ids)
        self._ids = (ctypes.c_int * max(1, len(ids)))(*ids)
        self._offs = (ctypes.c_int * max(1, len(offs)))(*offs)
        self._words = (ctypes.c_ulonglong * max(1, len(words)))(*words)
        self._failed = ctypes.c_int(-1)
        self._replay_fn = L.cdll.svla_replay_calls
        return self

    def replay(self):
        if getattr(self, "_n", None) != len(self.calls):
            self.compile()
        rc = self._replay_fn(self._n, self._ids, self._offs, self._words, ctypes.byref(self._failed))
        if rc != 0:
            raise RuntimeError(f"replayed {self.calls[self._failed.value][0].__name__} (call {self._failed.value}) failed with status {rc}")

    def replay_python(self):
        """The same sequence issued call by call from Python (the pre-round-3 path; kept for A/B: tools/replay_probe.py)."""
        for fn, a in self.calls:
            rc = fn(*a)
            if rc != 0:
                raise RuntimeError(f"replayed {fn.__name__} failed with status {rc}")


class GroupedPlans:
    """The recorded single-step sequences of the three towers (same entry points in the same order on the same shapes, different weights and
    buffers) replayed as ONE dependency chain of tower-grouped launches (svla_replay_calls_grouped, include/svla.h: call i of every tower inside
    one launch-group capture -- blockIdx.z / workgroup_id_z selects the tower's argument block) on ``stream`` instead of three chains on three
    streams: three times the workgroups per dispatch, a third of the dispatches, and no tower waiting behind another tower's chip-filling
    kernel.  Bit-identical to replaying the plans one by one (tests/test_grouped_gpu.py).  ``GroupedPlans.compatible(plans)`` says whether the
    sequences line up (if not -- e.g. towers with different critic heads -- the caller keeps the three-stream replay)."""

    def __init__(self, plans, stream: int):
        assert self.compatible(plans)
        L = lib()
        self.plans = list(plans)           # keeps the recorded tensors alive
        ref = plans[0]
        ids, offs, words = [], [], [[] for _ in plans]
        for i, (fn, a) in enumerate(ref.calls):
            name = fn.__name__
            decl = L.decls[name]
            ids.append(L.fn_ids[name])
            offs.append(len(words[0]))
            for m, pl in enumerate(plans):
                am = pl.calls[i][1]
                assert len(decl) == len(am), name
                for (an, ct), v in zip(decl, am):
                    if an == "stream":     # every launch of the group goes to the one stream the group is replayed on
                        words[m].append(int(stream) & 0xFFFFFFFFFFFFFFFF)
                    elif ct is ctypes.c_float:
                        words[m].append(struct.unpack("<I", struct.pack("<f", float(v)))[0])
                    elif ct is ctypes.c_double:
                        words[m].append(struct.unpack("<Q", struct.pack("<d", float(v)))[0])
                    elif v is None:
                        words[m].append(0)
                    elif isinstance(v, ctypes.c_void_p):
                        words[m].append(int(v.value or 0))
                    else:
                        words[m].append(int(v) & 0xFFFFFFFFFFFFFFFF)
        self._n, self._members = len(ids), len(plans)
        self._stream = ctypes.c_void_p(int(stream))
        self._ids = (ctypes.c_int * max(1, len(ids)))(*ids)
        self._offs = (ctypes.c_int * max(1, len(offs)))(*offs)
        self._words = [(ctypes.c_ulonglong * max(1, len(w)))(*w) for w in words]
        self._argv = (ctypes.POINTER(ctypes.c_ulonglong) * len(plans))(*[ctypes.cast(w, ctypes.POINTER(ctypes.c_ulonglong)) for w in self._words])
        self._failed = ctypes.c_int(-1)
        self._fn = L.cdll.svla_replay_calls_grouped

    @staticmethod
    def compatible(plans) -> bool:
        if len(plans) < 2 or len(plans) > 3:
            return False
        names = [[fn.__name__ for fn, _ in pl.calls] for pl in plans]
        return all(n == names[0] for n in names[1:]) and all(len(a) == len(b) for pl in plans[1:] for (_, a), (_, b) in zip(plans[0].calls, pl.calls))

    def replay(self):
        rc = self._fn(self._n, self._members, self._ids, self._offs, self._argv, self._stream, ctypes.byref(self._failed))
        if rc != 0:
            raise RuntimeError(f"grouped replay of {self.plans[0].calls[self._failed.value][0].__name__} (call {self._failed.value}) failed with status {rc}")


def group_stats():
    """(launches issued grouped, launches issued singly) by this thread's launch-group captures since the last call"""
    g, s1 = ctypes.c_long(0), ctypes.c_long(0)
    lib().cdll.svla_group_stats(ctypes.byref(g), ctypes.byref(s1))
    return g.value, s1.value


_REC: Optional["LaunchPlan"] = None


def _p(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if _REC is not None:
        _REC.keep.append(t)
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream():
    # the current HIP stream handle of the current device; the raw getter avoids building a torch.cuda.Stream object per launch
    # (2.8 us -> 0.3 us on the launch-bound acting path)
    if _raw_stream is not None:
        return _raw_stream(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def _chk(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise ValueError(f"{name} must be a CUDA/HIP tensor (no CPU fallback)")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


# ------------------------------------------------------------------------------------------------ rollout / losses
def gae_scan(rewards, costs, values, c_values, masks, next_v, next_cv, gamma=0.99, tau=0.95):
    """All [T,B] fp32 contiguous (masks [T+1,B]); returns (ret, adv, c_ret, c_adv) each [T,B]."""
    T, B = rewards.shape[:2]
    for n, t in (("rewards", rewards), ("costs", costs), ("values", values), ("c_values", c_values), ("masks", masks)):
        _chk(t, F32, n)
        assert t.is_contiguous()
    assert masks.shape[0] == T + 1
    out = [torch.empty(T, B, device=rewards.device, dtype=F32) for _ in range(4)]
    lib().call("svla_gae_scan_f32", _p(rewards), _p(costs), _p(values), _p(c_values), _p(masks), _p(next_v), _p(next_cv),
               float(gamma), float(tau), T, B, *[_p(o) for o in out], _stream())
    return tuple(out)


def ppo_lag_loss_fwd_bwd(logits, values, actions, old_logp, adv, c_adv, returns, old_values, lam, clip, value_coef,
                         action_w, ent_coef, use_clipped_value, inv_n, sums=None):
    """logits [R,A] f32, everything else [R]; returns (sums[3] double, dlogits [R,A], dvalues [R])."""
    R, A = logits.shape
    _chk(logits, F32, "logits")
    _chk(actions, torch.int64, "actions")
    dlogits = torch.empty_like(logits)
    dvalues = torch.empty(R, device=logits.device, dtype=F32)
    if sums is None:
        sums = torch.zeros(3, device=logits.device, dtype=torch.float64)
    lib().call("svla_ppo_lag_loss_fwd_bwd_f32", _p(logits), _p(values), _p(actions), _p(old_logp), _p(adv), _p(c_adv),
               _p(returns), _p(old_values), R, A, float(lam), float(clip), float(value_coef), float(action_w),
               float(ent_coef), int(bool(use_clipped_value)), float(inv_n), _p(dlogits), _p(dvalues), _p(sums), _stream())
    return sums, dlogits, dvalues


def value_mse_fwd_bwd(values, returns, coef, inv_n, sums=None, old_values=None, clip=0.0):
    """``old_values`` given: the clipped value loss (max of the plain and the clipped squared error)."""
    R = values.numel()
    dvalues = torch.empty(R, device=values.device, dtype=F32)
    if sums is None:
        sums = torch.zeros(1, device=values.device, dtype=torch.float64)
    lib().call("svla_value_mse_fwd_bwd_f32", _p(values), _p(returns), _p(old_values), float(clip), R, float(coef), float(inv_n),
               _p(dvalues), _p(sums), _stream())
    return sums, dvalues


def hlgauss_fwd_bwd(logits, target=None, dvalue=None, vmin=-5.0, vmax=15.0, sigma=0.15, coef=1.0, inv_n=1.0, want_values=True,
                    want_grad=True, sums=None):
    """logits [R, NB] fp32 -> (values [R] or None, dlogits [R, NB] or None, sums[1] double)."""
    _chk(logits, F32, "logits")
    R, NB = logits.shape
    values = torch.empty(R, device=logits.device, dtype=F32) if want_values else None
    dlogits = torch.empty_like(logits) if want_grad else None
    if sums is None and target is not None:
        sums = torch.zeros(1, device=logits.device, dtype=torch.float64)
    lib().call("svla_hlgauss_fwd_bwd_f32", _p(logits), _p(target), _p(dvalue), R, NB, float(vmin), float(vmax), float(sigma), float(coef),
               float(inv_n), _p(values), _p(dlogits), _p(sums), _stream())
    return values, dlogits, sums


def gemm_f32(A, B, M, N, K, sa=None, sb=None, bias=None, act=ACT_NONE, residual=None, mask=None, out=None, accumulate=False, alpha=1.0,
             drop=None, ldc=None, ldr=None, ldm=None):
    """fp32 strided GEMM: out[M,N] (+)= epi(alpha * A(m,k) B(n,k)); ``sa`` = (row stride, k stride) of A (default: [M,K] row-major),
    ``sb`` = (n stride, k stride) of B (default: [N,K] row-major, i.e. ``x @ W.T``)."""
    _chk(A, F32, "A"); _chk(B, F32, "B")
    sa = sa or (K, 1)
    sb = sb or (K, 1)
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=F32)
    lib().call("svla_gemm_f32", _p(A), int(sa[0]), int(sa[1]), _p(B), int(sb[0]), int(sb[1]), _p(bias), _p(residual), int(ldr or N),
               _p(mask), int(ldm or N), _p(out), int(ldc or N), M, N, K, int(act), int(bool(accumulate)), float(alpha), _d(drop), _stream())
    return out


def colsum_f32(X, out, M, N, row_stride=1, ldx=None):
    lib().call("svla_colsum_f32", _p(X), int(ldx or N), M, N, int(row_stride), _p(out), _stream())


def small_linear_fwd(x, W, bias, T=0, B=0):
    rows, D = x.shape
    N = W.shape[0]
    out = torch.empty(rows, N, device=x.device, dtype=F32)
    lib().call("svla_small_linear_fwd_f32", _p(x), _p(W), _p(bias), rows, N, D, T, B, _p(out), _stream())
    return out


def small_linear_bwd(x, W, dout, dx, dW, db, T=0, B=0, accumulate_dx=False):
    rows, D = x.shape
    N = W.shape[0]
    lib().call("svla_small_linear_bwd_f32", _p(x), _p(W), _p(dout), rows, N, D, T, B, int(accumulate_dx), _p(dx), _p(dW),
               _p(db), _stream())


# ------------------------------------------------------------------------------------------------ norms
def norm_fwd(x, gamma, beta, eps, rows, D=512, rms=False, relu=False, tok=None, tok_group=0, y=None,
             xmap=(0, 0, 0), ymap=(0, 0, 0), save_stats=True):
    f32 = x.dtype == F32            # fp32 verification mode: the same kernel instantiated on float rows
    if not f32:
        _chk(x, BF16, "x")
    if y is None:
        y = torch.empty(rows, D, device=x.device, dtype=x.dtype)
    mean = torch.empty(rows, device=x.device, dtype=F32) if (save_stats and not rms) else None
    rstd = torch.empty(rows, device=x.device, dtype=F32) if save_stats else None
    lib().call("svla_norm_fwd_f32" if f32 else "svla_norm_fwd_bf16", _p(x), *xmap, _p(gamma), _p(beta), float(eps), rows, D, int(rms), int(relu), _p(tok),
               int(tok_group), _p(y), *ymap, _p(mean), _p(rstd), _stream())
    return y, mean, rstd


def norm_bwd(dy, x, gamma, beta, mean, rstd, rows, dgamma, dbeta, D=512, rms=False, relu=False, dtok=None, tok_group=0,
             dx=None, dymap=(0, 0, 0), xmap=(0, 0, 0), dxmap=(0, 0, 0), dres=None, dx_drop=None, drop=None):
    """``dx_drop`` (optional [rows, D] output) = dx with the keep-mask / scale of dropout site ``drop`` applied: the gradient of
    the sub-layer output that was dropped out before being added to the residual stream."""
    f32 = dy.dtype == F32
    if not f32:
        _chk(dy, BF16, "dy")
    if dx is None:
        dx = torch.empty(rows, D, device=x.device, dtype=dy.dtype)
    lib().call("svla_norm_bwd_f32" if f32 else "svla_norm_bwd_bf16", _p(dy), *dymap, _p(x), *xmap, _p(gamma), _p(beta), _p(mean), _p(rstd), rows, D, int(rms),
               int(relu), int(tok_group), _p(dres), _p(dx), *dxmap, _p(dgamma), _p(dbeta), _p(dtok), _p(dx_drop), _d(drop), _stream())
    return dx


# ------------------------------------------------------------------------------------------------ dropout descriptor
class _SvlaDropout(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint), ("stream", ctypes.c_uint), ("p", ctypes.c_float), ("row_mult", ctypes.c_int),
                ("seed_dev", ctypes.c_void_p)]


class Dropout:
    """``svla_dropout`` of include/svla.h: one dropout site of one forward pass (seed = the pass, stream = the site)."""

    def __init__(self, seed: int, stream: int, p: float, row_mult: int = 1, seed_dev=None):
        """``seed_dev``: 1-element int32/uint32 device tensor holding the pass seed (read when the kernel starts; for captured graphs)."""
        self.seed_dev = seed_dev          # keep the tensor alive
        self.c = _SvlaDropout(seed & 0xFFFFFFFF, stream & 0xFFFFFFFF, float(p), int(row_mult), None if seed_dev is None else seed_dev.data_ptr())

    def with_row_mult(self, row_mult: int) -> "Dropout":
        return Dropout(self.c.seed, self.c.stream, self.c.p, row_mult, self.seed_dev)

    @property
    def scale(self) -> float:
        return 1.0 / (1.0 - self.c.p)


def _d(drop):
    if drop is None or drop.c.p <= 0:
        return None
    if _REC is not None:
        _REC.keep.append(drop)
    return ctypes.cast(ctypes.pointer(drop.c), ctypes.c_void_p)


def dropout_(x, drop):
    """In-place dropout of a contiguous [rows, N] bf16 tensor (element index = flat index)."""
    if drop is None:
        return x
    if x.dtype != F32:
        _chk(x, BF16, "x")
    lib().call("svla_dropout_f32" if x.dtype == F32 else "svla_dropout_bf16", _p(x), x.numel() // x.shape[-1], x.shape[-1], _d(drop), _stream())
    return x


# ------------------------------------------------------------------------------------------------ GEMMs
def gemm_nt(A, B, M, N, K, bias=None, residual=None, relu_mask=None, act=ACT_NONE, out=None, out_f32=False, alpha=1.0,
            lda=None, ldb=None, ldc=None, ldr=None, ldm=None, relu_bits_out=None, relu_bits=None, drop=None):
    """out[M,N] = epi(alpha * A[M,K] @ B[N,K]^T).  A/B bf16; leading dims default to the last-dim stride of 2-D views.
    fp32 operands (verification mode) run the fp32 strided GEMM with the same epilogue (ReLU masks as tensors, no bit masks)."""
    lda = lda if lda is not None else A.stride(-2)
    ldb = ldb if ldb is not None else B.stride(-2)
    if A.dtype == F32:
        if relu_bits is not None or relu_bits_out is not None:
            raise ValueError("fp32 mode keeps the ReLU mask as a tensor (relu_mask=), not as bits")
        if out is None:
            out = torch.empty(M, N, device=A.device, dtype=F32)
        return gemm_f32(A, B, M, N, K, sa=(lda, 1), sb=(ldb, 1), bias=bias, act=act, residual=residual, mask=relu_mask, out=out, alpha=alpha,
                        drop=drop, ldc=ldc if ldc is not None else out.stride(-2),
                        ldr=ldr if ldr is not None else (residual.stride(-2) if residual is not None else 0),
                        ldm=ldm if ldm is not None else (relu_mask.stride(-2) if relu_mask is not None else 0))
    _chk(A, BF16, "A")
    _chk(B, BF16, "B")
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=F32 if out_f32 else BF16)
    ldc = ldc if ldc is not None else out.stride(-2)
    ldr = ldr if ldr is not None else (residual.stride(-2) if residual is not None else 0)
    ldm = ldm if ldm is not None else (relu_mask.stride(-2) if relu_mask is not None else 0)
    lib().call("svla_gemm_nt_bf16", _p(A), lda, _p(B), ldb, _p(bias), _p(residual), ldr, _p(relu_mask), ldm, _p(out), ldc, M, N, K,
               int(act), int(out_f32), float(alpha), _p(relu_bits_out), _p(relu_bits), _d(drop), _stream())
    return out


def gemm_nt_rmsa(A, Bg, M, N, K, eps, bias=None, residual=None, act=ACT_NONE, out=None, drop=None):
    """out[M, N] = epi(RMSNorm(A) @ W^T) with the norm's gamma already folded into ``Bg`` = W * gamma[None, :] (bf16): small-M path (128-tile kernel), the row
    statistics come out of the GEMM's own A fragments -- no separate norm launch."""
    _chk(A, BF16, "A")
    _chk(Bg, BF16, "B")
    if out is None:
        out = torch.empty(M, N, device=A.device, dtype=BF16)
    lib().call("svla_gemm_nt_rmsa_bf16", _p(A), A.stride(-2), _p(Bg), Bg.stride(-2), _p(bias), _p(residual), residual.stride(-2) if residual is not None else 0,
               _p(out), out.stride(-2), M, N, K, int(act), float(eps), _d(drop), _stream())
    return out


def relu_bits_bytes(M: int, N: int) -> int:
    """Size of the opaque ReLU sign-bit buffer of gemm_nt(..., relu_bits_out=) (SVLA_RELU_BITS_BYTES in svla.h)."""
    return ((M + 31) // 32) * 32 * (N // 8)


def gemm_force_small_tile(on):
    """0 / False: normal dispatch; 1 / True: force the 128-tile GEMM kernels; 2: force the 256-tile kernels (the ones that carry the update)
    wherever their shape constraints hold, whatever the problem size (tests at golden-fixture sizes)."""
    lib().call("svla_gemm_force_small_tile", int(on))


def gemm_last_kernel():
    """(kernel name, (M, N, K) as dispatched) of the last big-GEMM launch of this process (svla_gemm_last_kernel): kernel-choice tests."""
    import ctypes

    buf = ctypes.create_string_buffer(96)
    mnk = (ctypes.c_int * 3)()
    lib().call("svla_gemm_last_kernel", ctypes.cast(buf, ctypes.c_void_p), 96, ctypes.cast(mnk, ctypes.c_void_p))
    return buf.value.decode(), tuple(mnk)


def attn_bwd_two_pass(on) -> None:
    """True: the dQ + dK/dV kernel pair instead of the single-pass attention backward (A/B and tests)."""
    lib().call("svla_attn_bwd_two_pass", int(bool(on)))


def gemm_tn_acc(dY, X, dW, M, N, K, ldy=None, ldx=None, ldw=None, db=None):
    """dW[N,K] (fp32) += dY[M,N]^T @ X[M,K];  optional fused bias gradient db[N] += dY.sum(0)."""
    if dY.dtype == F32:
        ldy = ldy if ldy is not None else dY.stride(-2)
        ldx = ldx if ldx is not None else X.stride(-2)
        gemm_f32(dY, X, N, K, M, sa=(1, ldy), sb=(1, ldx), out=dW, accumulate=True, ldc=ldw if ldw is not None else dW.stride(-2))
        if db is not None:
            colsum_f32(dY, db, M, N, ldx=ldy)
        return
    _chk(dY, BF16, "dY")
    _chk(X, BF16, "X")
    _chk(dW, F32, "dW")
    lib().call("svla_gemm_tn_f32acc", _p(dY), ldy if ldy is not None else dY.stride(-2), _p(X),
               ldx if ldx is not None else X.stride(-2), _p(dW), ldw if ldw is not None else dW.stride(-2), _p(db), M, N, K, _stream())


def colsum_acc(dY, db, M, N, ldy=None, row_stride=1):
    if dY.dtype == F32:
        return colsum_f32(dY, db, M, N, row_stride=row_stride, ldx=ldy if ldy is not None else dY.stride(-2))
    lib().call("svla_colsum_bf16", _p(dY), ldy if ldy is not None else dY.stride(-2), M, N, row_stride, _p(db), _stream())


# ------------------------------------------------------------------------------------------------ attention
def _f32_cols(t, W):
    """fp32 contiguous copy of the first W columns of a 2-D (possibly column-sliced) activation view"""
    return t[:, :W].float().contiguous()


def attn_fwd(q, k, v, ld, rows, S, H, scale, out=None, ldo=None, mask_mode=MASK_NONE, traj=None, bias=None, kvalid=None,
             save_lse=True, Sq=0, ldq=0, kv_rows=0, drop=None, head_dim=64):
    """q/k/v: bf16 views whose element (token, h*64+d) sits at token*ld + h*64 + d.  Sq > 0: only the first Sq queries of
    every row (q then holds Sq rows per batch row with row stride ldq).

    ``head_dim`` != 64 (the two imitation-learning presets with TransformerConfig(n, 768, 8): heads of 96): the MFMA kernels are built for 64-wide heads;
    those presets take the fp32 attention kernels (same masks, same dropout counters) through fp32 copies of the operands -- a slow path, not a tuned one."""
    nq = Sq if Sq > 0 else S
    W = H * head_dim
    if head_dim != 64 and q.dtype != F32:
        q32, k32, v32 = _f32_cols(q, W), _f32_cols(k, W), _f32_cols(v, W)
        o32, lse = attn_fwd(q32, k32, v32, W, rows, S, H, scale, mask_mode=mask_mode, traj=traj, bias=bias, kvalid=kvalid, save_lse=save_lse, Sq=Sq,
                            ldq=W if Sq > 0 else 0, kv_rows=kv_rows, drop=drop, head_dim=head_dim)
        if out is None:
            return o32.to(q.dtype), lse
        out[:, :W].copy_(o32)
        return out, lse
    if out is None:
        out = torch.empty(rows * nq, W, device=q.device, dtype=q.dtype)
    ldo = ldo if ldo is not None else out.stride(-2)
    lse = torch.empty(rows, H, nq, device=q.device, dtype=F32) if save_lse else None
    lib().call("svla_attn_fwd_f32" if q.dtype == F32 else "svla_attn_fwd_bf16", _p(q), _p(k), _p(v), ld, _p(out), ldo, _p(lse), rows, S, H, int(head_dim), float(scale), mask_mode,
               _p(traj), _p(bias), _p(kvalid), int(Sq), int(ldq), int(kv_rows), _d(drop), _stream())
    return out, lse


def attn_bwd(q, k, v, ld, o, ldo, lse, do, lddo, dq, dk, dv, ldd, rows, S, H, scale, mask_mode=MASK_NONE, traj=None,
             bias=None, kvalid=None, Sq=0, ldq=0, lddq=0, d_ws=None, drop=None, head_dim=64):
    if head_dim != 64 and q.dtype != F32:       # heads of 96: fp32 kernels through fp32 copies (see attn_fwd)
        W = H * head_dim
        q32, k32, v32, o32, do32 = (_f32_cols(t, W) for t in (q, k, v, o, do))
        dq32, dk32, dv32 = torch.empty_like(q32), torch.empty_like(k32), torch.empty_like(v32)
        attn_bwd(q32, k32, v32, W, o32, W, lse, do32, W, dq32, dk32, dv32, W, rows, S, H, scale, mask_mode=mask_mode, traj=traj, kvalid=kvalid, Sq=Sq,
                 ldq=W if Sq > 0 else 0, lddq=W if Sq > 0 else 0, drop=drop, head_dim=head_dim)
        dq[:, :W].copy_(dq32); dk[:, :W].copy_(dk32); dv[:, :W].copy_(dv32)
        return
    if q.dtype == F32:
        lib().call("svla_attn_bwd_f32", _p(q), _p(k), _p(v), ld, _p(o), ldo, _p(lse), _p(do), lddo, _p(dq), _p(dk), _p(dv), ldd, rows, S, H, int(head_dim),
                   float(scale), mask_mode, _p(traj), _p(kvalid), int(Sq), int(ldq), int(lddq), _d(drop), _stream())
        return
    if d_ws is None:   # [rows, H, Sq] fp32 workspace: rowsum(dO * O), handed from the dQ kernel to the dK/dV kernel
        d_ws = torch.empty(rows * H * (Sq or S), device=q.device, dtype=F32)
    lib().call("svla_attn_bwd_bf16", _p(q), _p(k), _p(v), ld, _p(o), ldo, _p(lse), _p(do), lddo, _p(dq), _p(dk), _p(dv), ldd,
               rows, S, H, 64, float(scale), mask_mode, _p(traj), _p(bias), _p(kvalid), int(Sq), int(ldq), int(lddq), _p(d_ws), _d(drop), _stream())


# ---- deterministic gradient accumulation ---------------------------------------------------------------------------------------
_DET_SLOTS = [None, None]


def det_config(slot: int, f32: Optional[torch.Tensor], shadow: Optional[torch.Tensor]) -> None:
    """Register (or with None, None: unregister) an int64 fixed-point shadow of an fp32 accumulation range (svla_det_config)."""
    if f32 is None:
        lib().call("svla_det_config", int(slot), None, None, 0)
        _DET_SLOTS[slot] = None
        return
    assert f32.dtype == F32 and shadow.dtype == torch.int64 and shadow.numel() == f32.numel() and f32.is_contiguous()
    lib().call("svla_det_config", int(slot), f32.data_ptr(), shadow.data_ptr(), f32.numel())
    _DET_SLOTS[slot] = (f32, shadow)


def det_active() -> bool:
    return _DET_SLOTS[0] is not None


def det_finalize(f32: torch.Tensor, shadow: torch.Tensor) -> None:
    """f32 += shadow * 2^-52; shadow = 0 (svla_det_finalize), on the current stream."""
    lib().call("svla_det_finalize", _p(f32), _p(shadow), f32.numel(), _stream())


def det_set_grid(frac_bits: int) -> None:
    """grid 2^-frac_bits of the deterministic shadows (svla_det_set_grid): partials below 2^(50 - frac_bits) enter them"""
    lib().call("svla_det_set_grid", int(frac_bits))


def det_grid_bits(n_total: int) -> int:
    """the engine's choice: 52 bits at >= 16 384 global rows, one bit fewer per halving of the minibatch (the partials are 1 / n_total-scaled), at least 36"""
    import math
    return max(36, min(52, 52 - max(0, math.ceil(math.log2(16384.0 / max(1, int(n_total)))))))


def det_bypass_count(reset: bool = True) -> int:
    """partials that had a registered shadow but took the plain fp32 atomic since the last reset (svla_det_bypass_count; synchronises the device)"""
    c = ctypes.c_ulonglong(0)
    lib().call("svla_det_bypass_count", ctypes.byref(c), int(bool(reset)))
    return int(c.value)


# ---- fp8 attention (BASELINE config 5) ---------------------------------------------------------------------------------------
def _fp8_sp(S: int) -> int:
    return 64 if S <= 64 else 128 if S <= 128 else 192 if S <= 192 else 256


class Fp8QKV:
    """e4m3 copies of one layer's Q, K, V head slices (token-major and reduction-major) + their scales (svla_attn_fp8_quant)."""
    __slots__ = ("ws", "scales", "rows", "S", "H")

    def __init__(self, ws, scales, rows, S, H):
        self.ws, self.scales, self.rows, self.S, self.H = ws, scales, rows, S, H


def attn_fp8_quant(qkv, ld, rows, S, H) -> Fp8QKV:
    _chk(qkv, BF16, "qkv")
    ws = torch.empty(rows * H * 6 * _fp8_sp(S) * 64, device=qkv.device, dtype=torch.uint8)
    scales = torch.empty(rows * H * 3, device=qkv.device, dtype=F32)
    lib().call("svla_attn_fp8_quant", _p(qkv), ld, rows, S, H, 64, _p(ws), _p(scales), _stream())
    return Fp8QKV(ws, scales, rows, S, H)


def attn_fp8_fwd(f8: Fp8QKV, scale, out=None, save_lse=True, drop=None):
    rows, S, H = f8.rows, f8.S, f8.H
    if out is None:
        out = torch.empty(rows * S, H * 64, device=f8.ws.device, dtype=BF16)
    lse = torch.empty(rows, H, S, device=f8.ws.device, dtype=F32) if save_lse else None
    lib().call("svla_attn_fp8_fwd", _p(f8.ws), _p(f8.scales), _p(out), out.stride(-2), _p(lse), rows, S, H, 64, float(scale), _d(drop), _stream())
    return out, lse


def attn_fp8_bwd(f8: Fp8QKV, o, lse, do, dq, dk, dv, ldd, scale, drop=None):
    rows, S, H = f8.rows, f8.S, f8.H
    sp = _fp8_sp(S)
    gws = torch.empty(rows * H * 2 * sp * 64, device=o.device, dtype=torch.uint8)
    gscale = torch.empty(rows * H, device=o.device, dtype=F32)
    dws = torch.empty(rows * H * sp, device=o.device, dtype=F32)
    lib().call("svla_attn_fp8_bwd", _p(f8.ws), _p(f8.scales), _p(o), o.stride(-2), _p(lse), _p(do), do.stride(-2), _p(gws), _p(gscale), _p(dws),
               _p(dq), _p(dk), _p(dv), ldd, rows, S, H, 64, float(scale), _d(drop), _stream())


# ------------------------------------------------------------------------------------------------ glue
def feat_to_tokens(feat, out, cam, ncam=2):
    """feat (R,C,7,12) or (R,C,P) fp32 -> out bf16 [R, ncam, P, C] slot ``cam``."""
    _chk(feat, F32, "feat")
    R, C = feat.shape[:2]
    P = feat[0, 0].numel()
    lib().call("svla_feat_to_tokens_f32" if out.dtype == F32 else "svla_feat_to_tokens", _p(feat), R, C, P, cam, ncam, _p(out), _stream())


def fusion_fill(fusion_token, text, gid, x0, R, S, L, text_off):
    lib().call("svla_fusion_fill_f32" if x0.dtype == F32 else "svla_fusion_fill", _p(fusion_token), _p(text), _p(gid), R, S, L, text_off, int(x0.shape[-1]), _p(x0), _stream())


def fusion_text_bwd(dx0, gid, T, B, S, L, text_off, dtext):
    lib().call("svla_fusion_text_bwd_f32" if dx0.dtype == F32 else "svla_fusion_text_bwd", _p(dx0), _p(gid), T, B, S, L, text_off, int(dtext.shape[-1]), _p(dtext), _stream())


def decoder_embed_fwd(xf, xf_row_stride, act_tab, hand_tab, div_term, prev_actions, masks, hand, time_step, T, B, out,
                      n_actions=20):
    lib().call("svla_decoder_embed_fwd_f32" if out.dtype == F32 else "svla_decoder_embed_fwd", _p(xf), xf_row_stride, _p(act_tab), _p(hand_tab), _p(div_term), _p(prev_actions),
               _p(masks), _p(hand), _p(time_step), T, B, n_actions, int(out.shape[-1]), _p(out), _stream())


def decoder_embed_bwd(dout, prev_actions, masks, hand, T, B, dxf, dxf_row_stride, d_act_tab, d_hand_tab, n_actions=20):
    lib().call("svla_decoder_embed_bwd_f32" if dout.dtype == F32 else "svla_decoder_embed_bwd", _p(dout), _p(prev_actions), _p(masks), _p(hand), T, B, n_actions, int(dout.shape[-1]), _p(dxf),
               dxf_row_stride, _p(d_act_tab), _p(d_hand_tab), _stream())


def rows_add(dst, dst_ld, src, src_ld, rows, D=512):
    lib().call("svla_rows_add_f32" if dst.dtype == F32 else "svla_rows_add_bf16", _p(dst), dst_ld, _p(src), src_ld, rows, D, _stream())


def zeros(*shape, device, dtype):
    """torch.zeros through the C ABI (allocation by torch, the fill is a recorded launch)."""
    t = torch.empty(*shape, device=device, dtype=dtype)
    lib().call("svla_zero_bytes", _p(t), t.numel() * t.element_size(), _stream())
    return t


def acting_stage(tok_src, tok_dst, pa_src, pa_dst, mask_src, mask_dst, hand_src, hand_dst, ts_src, ts_dst, ids_src, ids_dst, am_dst, am8_dst, kvalid_dst, t_dev,
                 B, L, max_steps, t, seeds, seed_inc):
    """svla_acting_stage: every input of a recorded acting step -> its static buffer, the T5 padding masks, the KV-window mask, the step counter and the seed bumps in ONE launch"""
    sp = [_p(s_) for s_ in seeds] + [None] * (3 - len(seeds))
    lib().call("svla_acting_stage", _p(tok_src), _p(tok_dst), tok_src.numel() * tok_src.element_size(), _p(pa_src), _p(pa_dst), _p(mask_src), _p(mask_dst), _p(hand_src), _p(hand_dst),
               _p(ts_src), _p(ts_dst), _p(ids_src), _p(ids_dst), _p(am_dst), _p(am8_dst), _p(kvalid_dst), _p(t_dev), int(B), int(L), int(max_steps), int(t), sp[0], sp[1], sp[2],
               int(seed_inc), _stream())


def kv_append(src, ld_src, cache, t_dev, B, width):
    """cache[b, *t_dev, :width] = src[b, :width] (cache: [rows, max_steps, width] bf16; t_dev: 0-d int64 device tensor)."""
    _chk(cache, BF16, "cache")
    lib().call("svla_kv_append_bf16", _p(src), int(ld_src), _p(cache), int(cache.shape[1]), int(width), _p(t_dev), int(B), _stream())


def swiglu_fwd(ab, M, Hd, out=None):
    if out is None:
        out = torch.empty(M, Hd, device=ab.device, dtype=ab.dtype)
    lib().call("svla_swiglu_fwd_f32" if ab.dtype == F32 else "svla_swiglu_fwd", _p(ab), M, Hd, _p(out), _stream())
    return out


def swiglu_bwd(ab, dg, M, Hd, dab=None):
    if dab is None:
        dab = torch.empty(M, 2 * Hd, device=ab.device, dtype=ab.dtype)
    lib().call("svla_swiglu_bwd_f32" if ab.dtype == F32 else "svla_swiglu_bwd", _p(ab), _p(dg), M, Hd, _p(dab), _stream())
    return dab


def row_hash(rows_u8):
    """rows_u8: contiguous 2-D byte view [n, row_bytes] (any dtype reinterpreted) -> int64 [n] content hashes."""
    n = rows_u8.shape[0]
    row_bytes = rows_u8[0].numel() * rows_u8.element_size()
    out = torch.empty(n, device=rows_u8.device, dtype=torch.int64)
    lib().call("svla_row_hash_u8", _p(rows_u8), n, row_bytes, _p(out), _stream())
    return out


def embed_gather(table, ids, out=None, dtype=BF16):
    n, D = ids.numel(), table.shape[1]
    if out is None:
        out = torch.empty(n, D, device=table.device, dtype=dtype)
    lib().call("svla_embed_gather_f32" if out.dtype == F32 else "svla_embed_gather_f32_bf16", _p(table), _p(ids), n, D, _p(out), _stream())
    return out


# ------------------------------------------------------------------------------------------------ optimiser
def sumsq(g, out):
    lib().call("svla_sumsq_f32", _p(g), g.numel(), _p(out), _stream())


def adam_step(p, g, m, v, p_bf16, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, gnorm_sq=None, max_norm=0.0, grad_scale=1.0,
              weight_decay=0.0):
    lib().call("svla_adam_step_f32", _p(p), _p(g), _p(m), _p(v), _p(p_bf16), p.numel(), float(lr), float(beta1), float(beta2),
               float(eps), int(step), _p(gnorm_sq), float(max_norm), float(grad_scale), float(weight_decay), _stream())


def ce_loss_fwd_bwd(logits, target, n_valid, dlogits, sums, ignore_index=-1):
    """logits [rows, A] fp32, target [rows] int64; n_valid: 1-element fp32 device tensor (# non-ignored rows)."""
    _chk(logits, F32, "logits")
    rows, A = logits.shape
    lib().call("svla_ce_loss_fwd_bwd_f32", _p(logits), _p(target), rows, A, int(ignore_index), _p(n_valid), _p(dlogits), _p(sums), _stream())


def cast_bf16(src, dst):
    if dst.dtype == F32:          # fp32 verification mode: activations stay fp32
        dst.copy_(src)
        return
    lib().call("svla_cast_f32_bf16", _p(src), _p(dst), src.numel(), _stream())


def transpose_cast_bf16(src, dst):
    rows, cols = src.shape
    if dst.dtype == F32:          # fp32 verification mode
        dst.copy_(src.t())
        return
    lib().call("svla_transpose_cast_f32_bf16", _p(src), rows, cols, _p(dst), _stream())


# ------------------------------------------------------------------------------------------------ frozen ViT preprocessor
def normalize_u8(x_u8, mean3, std3):
    y = torch.empty(x_u8.shape, device=x_u8.device, dtype=F32)
    m, s_ = [float(v) for v in mean3], [float(v) for v in std3]
    import ctypes
    lib().call("svla_normalize_u8_f32", _p(x_u8), x_u8.numel(), (ctypes.c_float * 3)(*m), (ctypes.c_float * 3)(*s_), _p(y), _stream())
    return y


def patchify_u8(frames_u8, mean3, std3, out, crop_x=3, P=14, gh=16, gw=27):
    """frames_u8 [B,H,W,3] -> out bf16 [B, gh*gw, KP] normalised im2col rows."""
    import ctypes
    B, H, W, _ = frames_u8.shape
    KP = out.shape[-1]
    lib().call("svla_patchify_u8_bf16", _p(frames_u8), B, H, W, crop_x, P, gh, gw, KP, (ctypes.c_float * 3)(*[float(v) for v in mean3]),
               (ctypes.c_float * 3)(*[float(v) for v in std3]), _p(out), _stream())


def vit_tokens(patch, cls, pos, B, NP, C, out):
    lib().call("svla_vit_tokens", _p(patch), _p(cls), _p(pos), B, NP, C, _p(out), _stream())


def adaptive_pool_tokens(x, B, skip, gh, gw, C, oh, ow, cam=0, ncam=1, tok_out=None, chw_out=None):
    lib().call("svla_adaptive_pool_tokens", _p(x), B, skip, gh, gw, C, oh, ow, cam, ncam, _p(tok_out), _p(chw_out), _stream())
